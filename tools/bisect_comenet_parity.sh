mkdir -p gpurun_out/bisect
for r in "" "comenet_group_pairs=0" "comenet_wide_single=0" "comenet_group_rows=0" "_wide_chain=0" ; do
  tag=$(echo "${r:-default}" | tr '=,' '__')
  DIG3D_SKIP_BOX_PROBE=1 DIG3D_ROUTES="$r" DIG3D_PARITY_REPORT=gpurun_out/bisect/$tag.json python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "comenet" > gpurun_out/bisect/$tag.log 2>&1
  echo "$tag rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bisect/$tag.json'))
for k,v in d.items():
    if 'out_per_molecule_rel' in v: print('  ',k, 'permol %.3e'%v['out_per_molecule_rel'], 'batchmax %.3e'%v['out_vs_oracle64'], 'gold permol %.3e'%v['gold32_per_molecule_rel'])
"
done
