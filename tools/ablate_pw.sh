#!/bin/bash
# Ablation builds of the persistent-weights dense kernel (csrc/dense.hip:k_linear_pw, compile-time PW_ABL bit mask: 1 no
# global stores, 2 no activation math, 4 no row loads, 8 no MFMA) linked against the other objects of the current build:
#   bash tools/ablate_pw.sh build          (here: hipcc cross-compiles; the .so files travel with the snapshot)
#   bash tools/ablate_pw.sh run M K N      (on the GPU box: tools/bench_dense.py M K N with every variant)
D=dig_amd/lib/abl; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
OBJS=$(ls dig_amd/lib/*.o | grep -v dense.o)
if [ "$1" = build ]; then
  for m in ${MASKS:-1 2 4 8 15}; do
    ( /opt/rocm/bin/hipcc $FLAGS -DPW_ABL=$m -c dig_amd/csrc/dense.hip -o $D/dense_pw$m.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $D/dense_pw$m.o -o $D/libdig3d_pw$m.so && echo built $m ) &
  done
  wait
else
  mkdir -p gpurun_out
  echo "--- full"; python tools/bench_dense.py $2 $3 $4 2>&1 | grep "^M="
  for f in $D/libdig3d_pw*.so; do echo "--- $f"; DIG3D_ABL_LIB=$f python tools/bench_dense.py $2 $3 $4 2>&1 | grep "^M="; done
fi
