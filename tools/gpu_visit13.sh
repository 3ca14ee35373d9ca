#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -p no:cacheprovider -x -k "linear or grouped or oracle_autograd or graphed_step_equals" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_c.log | cut -c1-200
for d in 3 1 3 1; do
DIG3D_SMALLM_WG_DIV=$d timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 div $d] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
done
