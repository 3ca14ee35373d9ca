#!/bin/bash
# MFMA / VALU busy counters of the HEAD kernels: the B = 32 step (eager) and the dense kernels at scale
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "MfmaUtil VALUBusy"; do
  i=$((i+1))
  rm -rf /tmp/pmc_s$i /tmp/pmc_d$i /tmp/pmc_e$i
  timeout 400 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pmc_s$i -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 5 --no-cpu-baseline --no-roofline --eager > $R/gpurun_out/pmc_s$i.log 2>&1; echo "pmc step [$pm] rc=$?"
  f=$(find /tmp/pmc_s$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/pmc_step_$i.csv
  timeout 400 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pmc_d$i -o p --output-format csv -- python $R/tools/bench_dense.py 262144 128 128 > $R/gpurun_out/pmc_d$i.log 2>&1; echo "pmc dense128 [$pm] rc=$?"
  f=$(find /tmp/pmc_d$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/pmc_dense128_$i.csv
  timeout 400 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pmc_e$i -o p --output-format csv -- python $R/tools/bench_dense.py 1048576 256 256 > $R/gpurun_out/pmc_e$i.log 2>&1; echo "pmc dense256 [$pm] rc=$?"
  f=$(find /tmp/pmc_e$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/pmc_dense256_$i.csv
done
ls -la $R/gpurun_out/pmc_*.csv
