#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4.log 2>&1; echo "[config4] $(tail -1 gpurun_out/bench_c4.log | cut -c60-200)"
cd /tmp; rm -rf $R/gpurun_out/prof_spherenet_qm9
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_spherenet_qm9 -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_spherenet_qm9.log 2>&1
find $R/gpurun_out/prof_spherenet_qm9 -name '*kernel_trace.csv' -delete
grep "k_reduce_many" $R/gpurun_out/prof_spherenet_qm9/bench_kernel_stats.csv | cut -c1-120
