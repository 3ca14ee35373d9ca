import os, sys
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo')
import torch
from dig_amd import _hip
if os.environ.get('DIG3D_ABL_LIB'):
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
import roofline_kernels as R
for n in ('edge_to_node', 'comenet_conv'):
    wl = R.WORKLOADS[n]()
    mean, mn = R.time_workload(wl, iters=30)
    print(f'{n}: {mean*1e3:.1f} us (min {mn*1e3:.1f}) -> {wl["bytes"]/mean/1e6:.0f} GB/s, err {wl["check"]():.2e}', flush=True)
    del wl; torch.cuda.empty_cache()
