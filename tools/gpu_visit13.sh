#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -x -k "run_api or comenet or pronet or flat_adam or optim" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_c.log | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5.log 2>&1; echo "[config5] $(tail -1 gpurun_out/bench_c5.log | cut -c60-200)"
done
