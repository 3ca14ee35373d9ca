"""GPU box: the ComENet aggregation kernels at the config-5 stress size and at the bench size, timed stand-alone
(tools/roofline_kernels.py workloads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import _hip
if os.environ.get('DIG3D_ABL_LIB'):          # an alternative build of the library (same-box A/B of a kernel change)
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
import roofline_kernels as R
for mol in (1024, 128):
    for n in ('comenet_conv', 'comenet_featconv'):
        wl = R.WORKLOADS[n](molecules=mol)
        mean, mn = R.time_workload(wl, iters=30)
        print(f'{n} molecules={mol}: {mean*1e3:.1f} us (min {mn*1e3:.1f}) -> {wl["bytes"]/mean/1e6:.0f} GB/s algorithmic, err {wl["check"]():.2e}', flush=True)
        del wl
        torch.cuda.empty_cache()
