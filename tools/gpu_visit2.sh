#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DIG3D_PARITY_REPORT=$R/gpurun_out/parity_report2.json
timeout 900 python -m pytest tests/test_gpu_diffops.py -q -p no:cacheprovider > gpurun_out/pytest_diffops.log 2>&1; echo "diffops rc=$?"
tail -40 gpurun_out/pytest_diffops.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -k "force or tiny or schnet or run_api" > gpurun_out/pytest_force.log 2>&1; echo "force rc=$?"
tail -40 gpurun_out/pytest_force.log | cut -c1-600
for w in dimenetpp_md17_force; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 gpurun_out/bench_$w.log | cut -c1-300
done
