#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log | cut -c1-600
for rep in 1 2; do
  timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_ab.log 2>&1; echo "[default] $(tail -1 gpurun_out/bench_ab.log | cut -c60-140)"
done
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$? $(tail -1 gpurun_out/bench_$w.log | cut -c50-150)"
done
cd /tmp; rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_bench -name '*kernel_trace.csv' -delete
