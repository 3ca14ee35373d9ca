#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "flat_adam" > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_quick.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "run_api or graphed or hip_graph" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -E "metric|Error" | cut -c1-200
DIG3D_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>&1 | grep -E "metric|Error" | cut -c1-200
