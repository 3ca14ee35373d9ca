#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "linear_mfma" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_quick.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; grep metric gpurun_out/bench_graph.log | cut -c1-200 || tail -5 gpurun_out/bench_graph.log
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
