export DIG3D_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 DIG3D_SKIP_BOX_PROBE=1
mkdir -p gpurun_out
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 5 --windows 3 --no-roofline --no-cpu-baseline --no-through-loader > gpurun_out/bench_forced_dist.log 2>&1; echo "rc=$?"; grep '"metric"' gpurun_out/bench_forced_dist.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('value','ms_per_step','scaling','rccl_ranks','allreduce_ms','compute_ms_per_rank','captures_per_rank','work_balance_max_over_mean')})"
MASTER_PORT=29518 timeout 300 python bench.py --gpus 1 --strong --global-batch 32 --workload spherenet_oc20 --steps 5 --warmup 5 --windows 3 --no-roofline --no-cpu-baseline --no-through-loader > gpurun_out/bench_forced_dist_strong.log 2>&1; echo "rc=$?"; grep '"metric"' gpurun_out/bench_forced_dist_strong.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('value','ms_per_step','scaling','rccl_ranks','allreduce_ms','compute_ms_per_rank')}, d['config']['global_batch'])"
tail -3 gpurun_out/bench_forced_dist_strong.log | cut -c1-300
