#!/bin/bash
# full validation of the round's HEAD: parity suite, smoke, the default bench line (rooflines + PMC + CPU baseline), the
# other BASELINE configs, kernel-stat profiles of all five configs
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log | cut -c1-200
unset DIG3D_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-400
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$? $(tail -1 gpurun_out/bench_$w.log | cut -c50-170)"
done
cd /tmp
for w in spherenet_qm9 dimenetpp_md17_force schnet_qm9 spherenet_oc20 comenet_128; do
rm -rf $R/gpurun_out/prof_$w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o bench --output-format csv -- python $R/bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > $R/gpurun_out/prof_$w.log 2>&1; echo "prof $w rc=$? $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$w.log | tail -1)"
find $R/gpurun_out/prof_$w -name '*kernel_trace.csv' -delete
done
