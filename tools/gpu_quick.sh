#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -p no:cacheprovider -k "chain or graphed or default" > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_models.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; grep metric gpurun_out/bench_graph.log | cut -c1-200 || tail -5 gpurun_out/bench_graph.log
