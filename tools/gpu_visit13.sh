#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -p no:cacheprovider -x -k "triplet or oracle_autograd or graphed_step_equals" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_c.log | cut -c1-200
for cap in 768 256 1536; do
DIG3D_TRIP_BWD_BLOCKS=$cap timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 cap $cap] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
DIG3D_TRIP_BWD_BLOCKS=$cap timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4.log 2>&1; echo "[config4 cap $cap] $(tail -1 gpurun_out/bench_c4.log | cut -c60-200)"
done
