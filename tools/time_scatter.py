"""The judged scatter_add kernel (csrc/segment.hip:k_segsum_sorted) stand-alone: default launch and a rows-per-worker sweep,
HIP-event timed, with the float4 copy ceiling of the box next to it.   python tools/time_scatter.py [M] [C] [seglen]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402

from dig_amd import _hip  # noqa: E402
if os.environ.get('DIG3D_ABL_LIB'):          # an alternative build of the library (same-box A/B of a kernel change)
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
import roofline_kernels as R  # noqa: E402
from dig_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seglen = int(sys.argv[3]) if len(sys.argv) > 3 else 17
cal = R.calibration_copy()
ms, _ = R.time_workload(cal, iters=30)
print('library', os.path.basename(_hip.LIB_PATH))
print(f'copy ceiling {(cal["read_bytes"] + cal["write_bytes"]) / ms / 1e6:.0f} GB/s')
del cal
wl = R.wl_scatter_add(M, C, seglen)
ms, mn = R.time_workload(wl, iters=50)
print(f'default: {ms * 1e3:.1f} us mean / {mn * 1e3:.1f} min -> {wl["bytes"] / ms / 1e6:.0f} GB/s = {wl["bytes"] / ms / 1e6 / 8000:.3f} of 8 TB/s; '
      f'max err {wl["check"]():.2e}')
idx = R.sorted_index(M, seglen).cuda()
S = int(idx[-1]) + 1
src = torch.randn(M, C, device='cuda')
for L in (16, 32, 48, 64, 96, 128, 256):
    for mode in (3,):
        keep = {}

        def go(L=L, mode=mode):      # the result stays referenced like in the roofline workload: launches alternate between
            keep['out'] = ops.scatter(src, idx, dim=0, dim_size=S, assume_sorted=True, tuning=(L, mode))      # two output buffers
        w2 = dict(launch=go)
        ms, mn = R.time_workload(w2, iters=30)
        print(f'L={L:4d} mode={mode}: {ms * 1e3:7.1f} us -> {wl["bytes"] / ms / 1e6:.0f} GB/s')
ms, mn = R.time_workload(wl, iters=50, warmup=30)
print(f'default again (30 warm-up launches): {ms * 1e3:.1f} us mean / {mn * 1e3:.1f} min -> {wl["bytes"] / ms / 1e6:.0f} GB/s')
