# 1-rank RCCL group: what the data-parallel plumbing of a step costs with nothing to exchange (async vs sync all-reduce)
export DIG3D_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 DIG3D_SKIP_BOX_PROBE=1
mkdir -p gpurun_out
B="--gpus 1 --steps 20 --warmup 10 --windows 5 --no-roofline --no-cpu-baseline --no-through-loader"
p=29530
for mode in async sync; do
  for w in spherenet_qm9 spherenet_oc20; do
    p=$((p+1))
    MASTER_PORT=$p timeout 300 python bench.py $B --workload $w --allreduce $mode > gpurun_out/dist1_${mode}_$w.log 2>&1
    echo "[$mode $w] rc=$?"; grep '"metric"' gpurun_out/dist1_${mode}_$w.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('value','ms_per_step','ms_p10','ms_p90','allreduce_ms','compute_ms_per_rank')})"
  done
done
unset DIG3D_FORCE_DIST RANK WORLD_SIZE LOCAL_RANK
timeout 300 python bench.py --steps 20 --warmup 10 --windows 5 --no-roofline --no-cpu-baseline --no-through-loader > gpurun_out/dist1_none.log 2>&1
grep '"metric"' gpurun_out/dist1_none.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('[no process group]', {k:d.get(k) for k in ('value','ms_per_step')})"
