#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "comenet or ComENet or pronet" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log | cut -c1-300
timeout 300 python tools/time_conv.py 2>&1 | grep -v amdgpu.ids
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5.log 2>&1; echo "[config5] $(tail -1 gpurun_out/bench_c5.log | cut -c60-200)"
