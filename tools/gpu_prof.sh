#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_bench.log
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
