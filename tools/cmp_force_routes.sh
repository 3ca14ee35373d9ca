for r in "" "force_sbf_fused=0" "force_sbf_fused=0,force_wide2=0,force_front2=0"; do
tag=$(echo "${r:-default}" | tr '=,' '__')
DIG3D_SKIP_BOX_PROBE=1 DIG3D_ROUTES="$r" DIG3D_PARITY_REPORT=gpurun_out/par_$tag.json python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "test_model_matches and force or test_force_route_chain" 2>&1 | tail -2
python -c "
import json; d=json.load(open('gpurun_out/par_$tag.json'))
for k,v in d.items(): print('$tag',k, {a: '%.2e'%v[a] for a in ('force_vs_oracle64','force_vs_gold32','force_gold32_noise','grad_vs_oracle64_global','grad_vs_oracle64','out_vs_oracle64','worst_grad','force_rel') if a in v})
"
done
