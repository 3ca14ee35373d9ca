"""GPU box: GraphNorm forward / backward launches (csrc/norm.hip) at ComENet's config-5 shape — B graphs of `atoms` rows,
C channels:  python tools/time_norm.py [B atoms C]   (DIG3D_ABL_LIB=<alt .so> for a same-box A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import _hip
from dig_amd._hip import call, ptr
if os.environ.get('DIG3D_ABL_LIB'):
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
B, atoms, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 256)
N = B * atoms
st = torch.cuda.current_stream().cuda_stream
x, gy = torch.randn(N, C, device='cuda'), torch.randn(N, C, device='cuda')
w, b, ms = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda'), torch.rand(C, device='cuda') + 0.5
gptr = (torch.arange(B + 1, device='cuda') * atoms).int()
y, gx = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(B, C, device='cuda'), torch.empty(B, C, device='cuda')
part, gp = torch.empty(B * 3 * C, device='cuda'), torch.empty(3 * C, device='cuda')

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3

tf = timeit(lambda: call('dig3d_graphnorm_fwd', ptr(x), ptr(gptr), B, C, ptr(w), ptr(b), ptr(ms), 1e-5, ptr(y), ptr(mean), ptr(rstd), st))
tb = timeit(lambda: call('dig3d_graphnorm_bwd', ptr(gy), ptr(x), ptr(gptr), B, C, ptr(w), ptr(ms), ptr(mean), ptr(rstd), ptr(gx), ptr(part), ptr(gp), st))
gb = N * C * 4 / 1e9
print(f'GRAPHNORM B={B} atoms={atoms} C={C}: fwd {tf:.1f} us ({2 * gb / tf * 1e3:.2f} TB/s)  bwd+colsum {tb:.1f} us ({3 * gb / tb * 1e3:.2f} TB/s)')
