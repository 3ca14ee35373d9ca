#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_integration.py -q -p no:cacheprovider -k "linear or integration or reference_network" > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_b.log | cut -c1-400
echo "--- persistent"; timeout 300 python tools/bench_dense.py 2>&1 | tail -4 | cut -c1-250
echo "--- DIG3D_NO_PERSISTENT"; DIG3D_NO_PERSISTENT=1 timeout 300 python tools/bench_dense.py 2>&1 | tail -4 | cut -c1-250
