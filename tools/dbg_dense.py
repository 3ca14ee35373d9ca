import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import _hip
from dig_amd._hip import call, ptr
st = torch.cuda.current_stream().cuda_stream
M, K, N = [int(v) for v in sys.argv[1:4]]
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda'); z = torch.empty(M, N, device='cuda')
for dbg in (0, 3, 4):
    _hip.query('dig3d_dense_debug', dbg)
    for _ in range(30):
        call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 1, ptr(y), ptr(z), st)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 1, ptr(y), ptr(z), s.cuda_stream)
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); e.record(); torch.cuda.synchronize()
    print(f'M={M} K={K} N={N} dbg={dbg}: {a.elapsed_time(e)/200*1e3:.2f} us per launch (200 back-to-back in a graph)', flush=True)
