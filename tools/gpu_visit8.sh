#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log | cut -c1-800
for v in "" "DIG3D_WGRAD_WORKERS=64" "DIG3D_WGRAD_WORKERS=96"; do
  env $v timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_ab.log 2>&1; echo "[$v] $(tail -1 gpurun_out/bench_ab.log | cut -c60-140)"
done
for v in "" "DIG3D_WGRAD_WORKERS=64"; do
  env $v timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dimenetpp_md17_force.log 2>&1; echo "cfg3 [$v] $(tail -1 gpurun_out/bench_dimenetpp_md17_force.log | cut -c50-150)"
done
cd /tmp; rm -rf $R/gpurun_out/prof_cfg3
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg3 -o cfg3 --output-format csv -- python $R/bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_cfg3.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_cfg3 -name '*kernel_trace.csv' -delete
