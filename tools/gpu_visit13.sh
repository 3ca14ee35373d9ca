#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -x -k "linear" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_c.log | cut -c1-300
for sh in "131072 256 256" "4200000 256 256" "262144 128 128"; do
echo "--- persistent $sh"; timeout 300 python tools/bench_dense.py $sh 2>&1 | tail -1 | cut -c1-250
echo "--- tiled $sh"; DIG3D_NO_PERSISTENT=1 timeout 300 python tools/bench_dense.py $sh 2>&1 | tail -1 | cut -c1-250
done
timeout 300 python tools/bench_dense.py --ablate 4200000 256 256 2>&1 | tail -1 | cut -c1-330
