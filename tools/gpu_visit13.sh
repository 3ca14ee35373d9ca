#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -p no:cacheprovider -x -k "chain or graphed_step_equals or oracle_autograd" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log | cut -c1-300
DIG3D_CHAIN_TILE=32 timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -x -k "chain" > gpurun_out/pytest_d.log 2>&1; echo "pytest(32) rc=$?"; tail -2 gpurun_out/pytest_d.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 auto(32)] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_CHAIN_TILE=64 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2n.log 2>&1; echo "[config2 tile 64] $(tail -1 gpurun_out/bench_c2n.log | cut -c60-200)"
done
