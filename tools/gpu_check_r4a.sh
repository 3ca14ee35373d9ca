#!/bin/bash
# round-4 visit A: NaN trace under the fence, degenerate-batch poison diagnosis, new triplet kernels, full suite, bench
out=gpurun_out/r4a
mkdir -p $out
export TMPDIR=/tmp
for m in hi lo; do
  DIG3D_EFENCE=$m timeout 300 python tools/efence/trace_nan.py SphereNet > $out/trace_$m.log 2>&1
done
DIG3D_EFENCE=hi timeout 300 python tools/efence/trace_nan.py SchNet > $out/trace_schnet.log 2>&1
DIG3D_EFENCE=hi timeout 300 python tools/efence/trace_nan.py ComENet > $out/trace_comenet.log 2>&1
timeout 300 python tools/diag_degenerate.py > $out/degenerate.log 2>&1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1
echo "pytest rc=$?"
timeout 600 python bench.py > $out/bench.log 2>&1
echo "bench rc=$?"
tail -3 $out/pytest.log
tail -c 1500 $out/bench.log
