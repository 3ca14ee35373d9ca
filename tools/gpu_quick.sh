#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "graphed or hip_graph or run_api" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; grep metric gpurun_out/bench_graph.log | cut -c1-200 || tail -5 gpurun_out/bench_graph.log
DIG3D_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_dist1.log 2>&1; echo "rc=$?"; grep metric gpurun_out/bench_dist1.log | cut -c1-220 || tail -8 gpurun_out/bench_dist1.log
timeout 300 python tools/dbg_nccl_graph.py 2>&1 | grep -E "ok|done|Error|error" | head -5
