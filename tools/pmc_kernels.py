"""Workload for the PMC passes (run under ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE``, one
counter per pass as MI355X_MICROARCH.md prescribes): the calibration copy of known size, then every roofline
workload named on the command line (default: all), REPS launches each.  tools/roofline_kernels.py:collect_pmc
drives the two passes and turns the counter CSVs into bytes per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import roofline_kernels as R

REPS = 6
names = sys.argv[1:] or list(R.WORKLOADS)
cal = R.calibration_copy()
for _ in range(REPS):
    cal['launch']()
torch.cuda.synchronize()
for n in names:
    wl = R.WORKLOADS[n]()
    for _ in range(REPS):
        wl['launch']()
    torch.cuda.synchronize()
    print('PMCWL', n, wl['kernel'], wl['bytes'], flush=True)
    del wl
    torch.cuda.empty_cache()
