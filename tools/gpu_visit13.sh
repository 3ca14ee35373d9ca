#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -k "pair_launch or graphed_step_equals or radial_bundle" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_c.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 pair] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_NO_PAIR=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2n.log 2>&1; echo "[config2 nopair] $(tail -1 gpurun_out/bench_c2n.log | cut -c60-200)"
done
timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3.log 2>&1; echo "[config3 pair] $(tail -1 gpurun_out/bench_c3.log | cut -c60-200)"
DIG3D_NO_PAIR=1 timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3n.log 2>&1; echo "[config3 nopair] $(tail -1 gpurun_out/bench_c3n.log | cut -c60-200)"
