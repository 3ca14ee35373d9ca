#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats. Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
