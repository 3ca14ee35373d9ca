#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "graphed or hip_graph or linear_mfma or closed or chain" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_models.log | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -E "metric|Error" | cut -c1-200
