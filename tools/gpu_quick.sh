#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "linear_mfma" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_quick.log | cut -c1-250
bash tools/gpu_prof.sh | cut -c1-150
