#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gspherenet.py tests/test_gpu_models.py -q -p no:cacheprovider -k "gspherenet or graphed or schnet" > gpurun_out/pytest_a.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_a.log | cut -c1-1500
for v in "" "DIG3D_NO_SMALL_M=1" "DIG3D_SMALL_M_FWD=1"; do
  for rep in 1 2; do
    env $v timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_ab.log 2>&1; echo "[$v] $(tail -1 gpurun_out/bench_ab.log | cut -c60-140)"
  done
done
timeout 300 python bench.py --workload schnet_qm9 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_schnet_qm9.log 2>&1; echo "schnet rc=$?"; tail -1 gpurun_out/bench_schnet_qm9.log | cut -c1-300
cd /tmp; rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_bench -name '*kernel_trace.csv' -delete
