#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_diffops.py tests/test_gpu_ops.py -q -p no:cacheprovider > gpurun_out/pytest_ops.log 2>&1; echo "ops rc=$?"
tail -30 gpurun_out/pytest_ops.log | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -k "comenet or run_api" > gpurun_out/pytest_comenet.log 2>&1; echo "comenet rc=$?"
tail -30 gpurun_out/pytest_comenet.log | cut -c1-500
cd /tmp; rm -rf $R/gpurun_out/prof_cfg3
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg3 -o cfg3 --output-format csv -- python $R/bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_cfg3.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_cfg3 -name '*kernel_trace.csv' -delete
tail -1 $R/gpurun_out/prof_cfg3.log | cut -c1-300
cd $R
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_comenet_128.log 2>&1; echo "comenet rc=$?"; tail -1 gpurun_out/bench_comenet_128.log | cut -c1-300
