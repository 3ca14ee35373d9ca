#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/dbg_nccl_graph.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "graphed or run_api" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log | cut -c1-300
