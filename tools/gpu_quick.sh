#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/sweep_wg.py 2>&1 | grep -v amdgpu.ids
