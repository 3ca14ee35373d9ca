#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for w in spherenet_oc20 comenet_128; do
rm -rf $R/gpurun_out/prof_$w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o bench --output-format csv -- python $R/bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_$w.log 2>&1; echo "prof $w rc=$? $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$w.log | tail -1)"
find $R/gpurun_out/prof_$w -name '*kernel_trace.csv' -delete
grep "embedding\|k_part_reduce" $R/gpurun_out/prof_$w/bench_kernel_stats.csv | cut -c1-50,150-260
done
cd $R
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 embed kernel] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_NO_EMBED_KERNEL=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2n.log 2>&1; echo "[config2 framework] $(tail -1 gpurun_out/bench_c2n.log | cut -c60-200)"
