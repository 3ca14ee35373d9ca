#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_b512
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b512 -o p --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --batch 512 > gpurun_out/prof_b512.log 2>&1; echo "prof rc=$?"
rm -f gpurun_out/prof_b512/p_kernel_trace.csv
