#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
export DIG3D_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/parity_report.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5.log 2>&1; echo "[config5 composed] $(tail -1 gpurun_out/bench_c5.log | cut -c60-200)"
DIG3D_NO_COMPOSE=1 timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5n.log 2>&1; echo "[config5 two-step] $(tail -1 gpurun_out/bench_c5n.log | cut -c60-200)"
done
