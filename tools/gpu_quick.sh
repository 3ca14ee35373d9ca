#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; grep metric gpurun_out/bench_graph.log | cut -c1-200 || tail -5 gpurun_out/bench_graph.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline --num-spherical 3 > gpurun_out/bench_graph3.log 2>&1; grep metric gpurun_out/bench_graph3.log | cut -c1-200
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-200
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
