"""GPU box: time the f32-MFMA dense kernels (csrc/dense.hip) against torch/hipBLASLt on the shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops, _hip
from dig_amd._hip import call, ptr
if os.environ.get('DIG3D_ABL_LIB'):          # an alternative build of the library (same-box A/B of a kernel change)
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

st = torch.cuda.current_stream().cuda_stream
SHAPES = [(8418, 128, 128), (8418, 384, 128), (8418, 128, 64), (8418, 64, 128), (600, 256, 256), (8418, 6, 128),
          (16384, 256, 256), (131072, 256, 256), (262144, 128, 128), (4200000, 256, 256)]
ABLATE = '--ablate' in sys.argv
if ABLATE:
    sys.argv.remove('--ablate')
if len(sys.argv) > 3:
    SHAPES = [tuple(int(v) for v in sys.argv[1:4])]
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    gy = torch.randn(M, N, device='cuda'); y = torch.empty(M, N, device='cuda'); z = torch.empty(M, N, device='cuda')
    gx = torch.empty(M, K, device='cuda')
    nb = max(_hip.query('dig3d_linear_wgrad_blocks', M), _hip.query('dig3d_linear_bwd_workers', M, K, N))
    part = torch.empty(nb * (N * K + N), device='cuda'); gwb = torch.empty(N * K + N, device='cuda')
    if ABLATE:     # which part of the forward costs what: optional pre-activation output, activation, residual; copy bandwidth
        r = torch.randn(M, N, device='cuda')
        t0 = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 1, ptr(y), ptr(z), st))
        t1 = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 1, ptr(y), None, st))
        t2 = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 0, ptr(y), None, st))
        t3 = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), ptr(r), M, K, N, 1, ptr(y), ptr(z), st))
        t4 = timeit(lambda: call('dig3d_linear_bwd_input', ptr(gy), None, ptr(w), M, K, N, 0, ptr(gx), None, st))
        t5 = timeit(lambda: y.copy_(gy))
        t6 = timeit(lambda: y.fill_(1.0))
        t7 = timeit(lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, w, b)))
        print(f'M={M} K={K} N={N}: fwd swish+Z {t0:.1f}us | swish {t1:.1f} | plain {t2:.1f} | swish+Z+res {t3:.1f} | '
              f'dgrad plain {t4:.1f} | copy [M,N] {t5:.1f}us ({2*M*N*4/t5/1e6:.2f} TB/s) fill {t6:.1f}us ({M*N*4/t6/1e6:.2f} TB/s) | torch linear+silu {t7:.1f}', flush=True)
        continue
    t_f = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 1, ptr(y), ptr(z), st))
    t_d = timeit(lambda: call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(w), M, K, N, 1, ptr(gx), None, st))
    t_w = timeit(lambda: call('dig3d_linear_bwd_weight', ptr(gy), ptr(z), ptr(x), M, K, N, 1, ptr(part), ptr(gwb), 1, st))
    t_b = timeit(lambda: call('dig3d_linear_bwd', ptr(gy), ptr(z), ptr(w), ptr(x), M, K, N, 1, ptr(gx), None, ptr(part), ptr(gwb), 1, st))
    # act = 3: Z already holds act'(z) (forward launched with act | 4): the backward kernels only multiply
    t_f5 = timeit(lambda: call('dig3d_linear_fwd', ptr(x), ptr(w), ptr(b), None, M, K, N, 5, ptr(y), ptr(z), st))
    t_d3 = timeit(lambda: call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(w), M, K, N, 3, ptr(gx), None, st))
    t_w3 = timeit(lambda: call('dig3d_linear_bwd_weight', ptr(gy), ptr(z), ptr(x), M, K, N, 3, ptr(part), ptr(gwb), 1, st))
    t_tf = timeit(lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, w, b)))
    t_td = timeit(lambda: gy @ w)
    t_tw = timeit(lambda: gy.t() @ x)
    fl = 2.0 * M * K * N
    print(f'M={M} K={K} N={N}: fwd {t_f:.1f}us ({fl/t_f/1e6:.1f} TF) dgrad {t_d:.1f}us ({fl/t_d/1e6:.1f} TF) wgrad+reduce {t_w:.1f}us both {t_b:.1f}us ({2*fl/t_b/1e6:.1f} TF) | '
          f'torch fwd+silu {t_tf:.1f} dgrad {t_td:.1f} wgrad {t_tw:.1f} | derivative kept: fwd {t_f5:.1f} dgrad {t_d3:.1f} wgrad {t_w3:.1f}', flush=True)
