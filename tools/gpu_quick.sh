#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for spin in 0 1; do
DIG3D_SPIN_WAIT=$spin timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spin=$spin', round(d['ms_per_step'],4))"
done; done
