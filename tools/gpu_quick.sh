#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -E "metric|Error" | cut -c1-200
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 2>&1 | grep -E "metric|Error" | cut -c1-190
done
