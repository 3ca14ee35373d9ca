#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -x -k "linear" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_c.log | cut -c1-300
echo "--- pw"; timeout 300 python tools/bench_dense.py 16384 256 256 2>&1 | tail -1 | cut -c1-250
echo "--- tiled"; DIG3D_NO_PERSISTENT=1 timeout 300 python tools/bench_dense.py 16384 256 256 2>&1 | tail -1 | cut -c1-250
for i in 1 2; do
timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5.log 2>&1; echo "[config5 pw at 1536 tiles] $(tail -1 gpurun_out/bench_c5.log | cut -c60-200)"
DIG3D_PW_MIN_TILES=100000000 timeout 400 python bench.py --workload comenet_128 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c5n.log 2>&1; echo "[config5 tiled] $(tail -1 gpurun_out/bench_c5n.log | cut -c60-200)"
done
