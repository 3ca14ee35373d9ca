#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_properties.py -x -q -p no:cacheprovider > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/pytest_quick.log | cut -c1-250
