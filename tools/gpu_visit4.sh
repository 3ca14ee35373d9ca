#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -45 gpurun_out/pytest_gpu.log | cut -c1-400
timeout 600 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-700
for w in dimenetpp_md17_force spherenet_oc20; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 gpurun_out/bench_$w.log | cut -c1-300
done
cd /tmp; rm -rf $R/gpurun_out/prof_cfg3
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg3 -o cfg3 --output-format csv -- python $R/bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_cfg3.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_cfg3 -name '*kernel_trace.csv' -delete
