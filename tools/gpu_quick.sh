#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "linear_mfma" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_quick.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log
timeout 300 python tools/bench_dense.py > gpurun_out/bench_dense.log 2>&1; echo "dense rc=$?"; cat gpurun_out/bench_dense.log
timeout 300 python tools/time_triplet.py > gpurun_out/time_triplet.log 2>&1; echo "time rc=$?"; cat gpurun_out/time_triplet.log
