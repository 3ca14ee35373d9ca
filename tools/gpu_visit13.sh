#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4.log 2>&1; echo "[config4 swizzle] $(tail -1 gpurun_out/bench_c4.log | cut -c60-200)"
DIG3D_NO_XCD_SWIZZLE=1 timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4n.log 2>&1; echo "[config4 natural] $(tail -1 gpurun_out/bench_c4n.log | cut -c60-200)"
done
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 swizzle] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_NO_XCD_SWIZZLE=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2n.log 2>&1; echo "[config2 natural] $(tail -1 gpurun_out/bench_c2n.log | cut -c60-200)"
