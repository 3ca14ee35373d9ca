"""GPU box: the deferred partial reduction (csrc/dense.hip:k_reduce_many) on the table of a SphereNet step — 44 layers of
128 x 128 (+ bias) with 11 partials each plus a few long tables:  python tools/time_reduce.py   (DIG3D_ABL_LIB=<alt .so>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import _hip
from dig_amd._hip import call, ptr
if os.environ.get('DIG3D_ABL_LIB'):
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
st = torch.cuda.current_stream().cuda_stream
spec = [(11, 128 * 128 + 128)] * 44 + [(256, 2 * 8 * 336)] * 4 + [(136, 6 * 128 + 128)] * 10 + [(38, 257)] * 5
parts = [torch.randn(nb * n, device='cuda') for nb, n in spec]
outs = [torch.empty(n, device='cuda') for _, n in spec]
c = len(spec)
PP, IA, LA = ctypes.c_void_p * c, ctypes.c_int * c, ctypes.c_int64 * c
cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
args = (cast(PP(*[ptr(p) for p in parts])), cast(IA(*[nb for nb, _ in spec])), cast(LA(*[n for _, n in spec])),
        cast(IA(*[n for _, n in spec])), cast(PP(*[ptr(o) for o in outs])), c, st)
f = lambda: call('dig3d_reduce_many', *args)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): f()
b.record(); torch.cuda.synchronize()
ref = sum(float(p.view(nb, n).sum(0).abs().sum()) for p, (nb, n) in zip(parts, spec))
got = sum(float(o.abs().sum()) for o in outs)
mb = sum(p.numel() for p in parts) * 4 / 1e6
print(f'REDUCE {c} tables, {mb:.1f} MB of partials: {a.elapsed_time(b) * 10:.1f} us per launch  (checksum {got:.6e} vs torch {ref:.6e})')
