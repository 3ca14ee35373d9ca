#!/bin/bash
# Ablation builds of csrc/chain.hip (compile-time CHAINR_ABL bit mask: 1 no global stores, 2 no activation math, 4 no
# next-weight loads, 8 no MFMA, 16 no residual / bias loads) linked against the other objects of the current build:
#   bash tools/ablate_chain.sh build        (here: hipcc cross-compiles; the .so files travel with the snapshot)
#   bash tools/ablate_chain.sh run M K0     (on the GPU box: graph-replayed timings of every variant)
D=dig_amd/lib/abl; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
OBJS=$(ls dig_amd/lib/*.o | grep -v chain.o)
if [ "$1" = build ]; then
  for m in ${MASKS:-1 2 4 8 16 31}; do
    /opt/rocm/bin/hipcc $FLAGS -DCHAINR_ABL=$m -c dig_amd/csrc/chain.hip -o $D/chain_$m.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $D/chain_$m.o -o $D/libdig3d_abl$m.so && echo built $m
  done
else
  mkdir -p gpurun_out
  python tools/bench_chain.py $2 $3 3 > gpurun_out/abl_base.log 2>&1; grep GRAPH gpurun_out/abl_base.log || tail -5 gpurun_out/abl_base.log
  for f in $D/libdig3d_abl*.so; do DIG3D_ABL_LIB=$f python tools/bench_chain.py $2 $3 3 2>&1 | grep GRAPH; done
fi
