#!/bin/bash
export TMPDIR=/tmp
for mb in 1 -2 -4; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --micro-batches=$mb 2>&1 | grep -E "metric|Error|error" | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('micro=$mb', round(d['ms_per_step'],3), 'ms', round(d['value']), 'mol/s')
except Exception: print('micro=$mb FAILED', t[-300:])"; done
