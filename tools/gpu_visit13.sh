#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
export DIG3D_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/parity_report.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py --workload spherenet_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3s.log 2>&1; echo "[spherenet force] $(tail -1 gpurun_out/bench_c3s.log | cut -c60-200)"
