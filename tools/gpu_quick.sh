#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "closed_matmul or linear_mfma" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_quick.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "force or reference_and_oracle" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_models.log | cut -c1-300
timeout 600 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 3 > gpurun_out/bench_force.log 2>&1; grep metric gpurun_out/bench_force.log | cut -c1-200 || tail -5 gpurun_out/bench_force.log
