#!/bin/bash
# round-2 GPU visit 1: full parity suite (new gradient assertions), bench with in-model rooflines + in-run PMC traffic,
# kernel-trace profile of the bench, the other BASELINE configs, the counter list + an MFMA/VALU-busy counter pass.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log | cut -c1-400
timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-3000
cd /tmp
rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-pmc > $R/gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
find $R/gpurun_out/prof_bench -name '*kernel_trace.csv' -delete
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "MfmaUtil VALUBusy"; do
  tag=$(echo $pm | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout 400 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pmc_$tag -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 5 --no-cpu-baseline --no-roofline --eager > $R/gpurun_out/pmc_$tag.log 2>&1; echo "pmc [$pm] rc=$?"
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_$tag.csv && ls -la $R/gpurun_out/pmc_$tag.csv
done
cd $R
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 gpurun_out/bench_$w.log | cut -c1-400
done
