import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8704
x = torch.randn(M, 64, device='cuda'); xji = torch.randn(M, 128, device='cuda'); x1 = torch.randn(M, 128, device='cuda')
W0 = torch.randn(128, 64, device='cuda') / 8
Ws = [torch.randn(128, 128, device='cuda') / 11 for _ in range(7)]
bs = [torch.randn(128, device='cuda') for _ in range(7)]
A = ops.ACT_SWISH
layers = [(W0, None, A, 1, xji, True), (Ws[0], bs[0], A, 0, None, False), (Ws[1], bs[1], A, 2, None, True), (Ws[2], bs[2], A, 1, x1, True),
          (Ws[3], bs[3], A, 0, None, False), (Ws[4], bs[4], A, 2, None, True), (Ws[5], bs[5], A, 0, None, False), (Ws[6], bs[6], A, 2, None, True)]
with torch.no_grad():
    for _ in range(5): y = ops.chain(x, layers)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(50): y = ops.chain(x, layers)
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); torch.cuda.synchronize()
print(f"M={M} dbg={os.environ.get('DIG3D_CHAIN_DBG','0')}: {a.elapsed_time(e)/50*1e3:.1f} us per 8-layer chain", flush=True)
