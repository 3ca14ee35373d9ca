#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "radial or oracle or graphed_step_equals or matches_reference" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 split] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_NO_RADIAL_SPLIT=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2n.log 2>&1; echo "[config2 one block per tile] $(tail -1 gpurun_out/bench_c2n.log | cut -c60-200)"
done
