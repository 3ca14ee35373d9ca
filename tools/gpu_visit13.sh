#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -x -k "comenet or pronet" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_c.log | cut -c1-200
cd /tmp; rm -rf $R/gpurun_out/prof_comenet_128
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comenet_128 -o bench --output-format csv -- python $R/bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_comenet_128.log 2>&1
find $R/gpurun_out/prof_comenet_128 -name '*kernel_trace.csv' -delete
grep "featconv" $R/gpurun_out/prof_comenet_128/bench_kernel_stats.csv | cut -c1-60,180-260
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_comenet_128.log | tail -1
