#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -k "force" > gpurun_out/pytest_force.log 2>&1; echo "force rc=$?"; tail -3 gpurun_out/pytest_force.log | cut -c1-300
timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dimenetpp_md17_force.log 2>&1; echo "cfg3 $(tail -1 gpurun_out/bench_dimenetpp_md17_force.log | cut -c50-150)"
echo "--- default"; timeout 300 python tools/bench_dense.py 2>&1 | tail -10 | cut -c1-230
echo "--- small always (+fwd)"; DIG3D_SMALL_M_ALWAYS=1 DIG3D_SMALL_M_FWD=1 DIG3D_SMALL_M_BOTH=1 timeout 300 python tools/bench_dense.py 2>&1 | tail -10 | cut -c1-230
