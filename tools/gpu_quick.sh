#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "hip_graph" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_models.log | cut -c1-250
