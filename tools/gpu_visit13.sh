#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -x -k "force" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_c.log | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3.log 2>&1; echo "[config3 fold] $(tail -1 gpurun_out/bench_c3.log | cut -c60-200)"
DIG3D_NO_FOLD_E2=1 timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3n.log 2>&1; echo "[config3 no fold] $(tail -1 gpurun_out/bench_c3n.log | cut -c60-200)"
done
