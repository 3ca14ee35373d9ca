#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "graphed or hip_graph" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_models.log | cut -c1-300
for mb in 1 2 4 1 2 4; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --micro-batches $mb 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('micro=$mb', round(d['ms_per_step'],3), 'ms', round(d['value']), 'mol/s')"; done
