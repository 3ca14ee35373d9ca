#!/bin/bash
# kernel-stat profiles of every BASELINE config on HEAD (graph replay where the model supports it)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for w in spherenet_qm9 dimenetpp_md17_force schnet_qm9 spherenet_oc20 comenet_128; do
rm -rf $R/gpurun_out/prof_$w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o bench --output-format csv -- python $R/bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_$w.log 2>&1; echo "prof $w rc=$? $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$w.log | tail -1)"
find $R/gpurun_out/prof_$w -name '*kernel_trace.csv' -delete
done
