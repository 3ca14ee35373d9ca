#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for i in 1 2; do
DIG3D_TRIP_XCD_SWIZZLE=1 timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4.log 2>&1; echo "[config4 trip_fwd swizzle] $(tail -1 gpurun_out/bench_c4.log | cut -c60-200)"
timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4n.log 2>&1; echo "[config4 natural] $(tail -1 gpurun_out/bench_c4n.log | cut -c60-200)"
done
