#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log | cut -c1-300
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/bench_$w.log 2>&1; echo "$w rc=$?"; grep metric gpurun_out/bench_$w.log | cut -c1-330 || tail -3 gpurun_out/bench_$w.log
done
