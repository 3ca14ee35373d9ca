#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "comenet or ComENet" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log | cut -c1-300
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5.log 2>&1; echo "[config5 featconv] $(tail -1 gpurun_out/bench_c5.log | cut -c60-200)"
DIG3D_NO_FEATCONV=1 timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5n.log 2>&1; echo "[config5 edge-weight tensor] $(tail -1 gpurun_out/bench_c5n.log | cut -c60-200)"
cd /tmp; rm -rf $R/gpurun_out/prof_comenet_128
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comenet_128 -o bench --output-format csv -- python $R/bench.py --workload comenet_128 --steps 5 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_comenet_128.log 2>&1
find $R/gpurun_out/prof_comenet_128 -name '*kernel_trace.csv' -delete

