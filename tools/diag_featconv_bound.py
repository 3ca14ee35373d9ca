"""GPU box: what bounds k_featconv?  The config-5-sized launch (128 molecules x 128 atoms x 32 neighbours, C = 256, K = 12) timed
with its real source rows, with every source row = the segment's own row (the gathers of a segment hit L1), and with all
source rows = row 0; and the same with the forward's segments reversed in launch order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import _hip
if os.environ.get('DIG3D_ABL_LIB'):
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
from dig_amd._hip import call, ptr
from dig_amd.graph import _stream
import roofline_kernels as R

for mol in (128, 1024):
    atoms, deg, C = 128, 32, 256
    N, E = mol * atoms, mol * atoms * deg
    for K in (12, 6):
        g = torch.Generator(device='cpu').manual_seed(3)
        pick = torch.rand(N, atoms, generator=g).argsort(1)[:, :deg].sort(1).values
        real = (pick + (torch.arange(N) // atoms * atoms).unsqueeze(1)).reshape(-1).to(torch.int32).cuda()
        own = torch.arange(N, dtype=torch.int32).repeat_interleave(deg).cuda()
        zero = torch.zeros(E, dtype=torch.int32).cuda()
        kptr = (torch.arange(N + 1, dtype=torch.int64) * deg).to(torch.int32).cuda()
        X = torch.randn(N, C, device='cuda'); F = torch.randn(E, K, device='cuda'); Wc = torch.randn(C, K, device='cuda')
        out = torch.empty(N, C, device='cuda')
        for name, src in (('real', real), ('own row', own), ('row 0', zero)):
            wl = dict(launch=lambda: call('dig3d_featconv', ptr(X), ptr(src), ptr(F), K, ptr(Wc), ptr(kptr), None, N, C, ptr(out), None, _stream()))
            mean, mn = R.time_workload(wl, iters=30)
            print(f'molecules={mol} K={K} sources={name}: {mean*1e3:.1f} us (min {mn*1e3:.1f})  gathers {E*C*4/mean/1e9:.2f} TB/s', flush=True)
