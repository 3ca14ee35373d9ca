#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "linear" > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_b.log | cut -c1-400
