#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cap in 256 512 256 512; do
DIG3D_CHAIN_WGRAD_BLOCKS=$cap timeout 400 python bench.py --workload spherenet_oc20 --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c4.log 2>&1; echo "[config4 chain wgrad blocks $cap] $(tail -1 gpurun_out/bench_c4.log | cut -c60-200)"
DIG3D_CHAIN_WGRAD_BLOCKS=$cap timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.log 2>&1; echo "[config2 $cap] $(tail -1 gpurun_out/bench_c2.log | cut -c60-200)"
done
