#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "linear_mfma or closed or flat_adam" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_quick.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -E "metric|Error" | cut -c1-200
