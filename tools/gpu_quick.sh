#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for bits in 2 4; do
DIG3D_BUCKET_BITS=$bits timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bits=$bits', d['ms_per_step'])"
done; done
