#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_properties.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_models.log | cut -c1-250
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep -E "metric|Error" | cut -c1-700
