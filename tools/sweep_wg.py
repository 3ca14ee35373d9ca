import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dig_amd.threedgraph.method as M
from dig_amd import _hip
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import make_batch, batch_to
b = batch_to(make_batch(32, 9, 29, 0.08, 5.0, seed=1), 'cuda')
for nw in (32, 48, 64, 96, 128, 192):
    _hip.call('dig3d_set_wgrad_workers', nw)
    torch.manual_seed(0)
    m = M.SphereNet(num_layers=4, hidden_channels=128).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, fused=True)
    st = GraphedStep(m)
    for _ in range(5):
        st(b); opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        st(b); opt.step()
    torch.cuda.synchronize()
    print(f'wgrad workers {nw}: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step', flush=True)
