#!/bin/bash
# fresh kernel-stat profiles of configs 2 and 3 (graph replay) + their bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2] $(tail -1 gpurun_out/bench_c2.log | cut -c1-200)"
timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3.log 2>&1; echo "[config3] $(tail -1 gpurun_out/bench_c3.log | cut -c1-200)"
cd /tmp
for w in spherenet_qm9 dimenetpp_md17_force; do
rm -rf $R/gpurun_out/prof_$w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o bench --output-format csv -- python $R/bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_$w.log 2>&1; echo "prof $w rc=$?"
find $R/gpurun_out/prof_$w -name '*kernel_trace.csv' -delete
done
