"""which parameter gradient of the degenerate batch (isolated atoms, no triplets) reads an unwritten slot:
torch.empty poisoned with NaN / INT_MAX, per model class and parameter."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import dig_amd.threedgraph.method as M
DEV = 'cuda'
pos = torch.tensor([[0., 0, 0], [50., 0, 0], [100., 0, 0], [100.9, 0, 0]], device=DEV)
b = SimpleNamespace(z=torch.tensor([1, 6, 8, 1], device=DEV), pos=pos, batch=torch.tensor([0, 1, 2, 2], device=DEV),
                    y=torch.zeros(3, device=DEV), node_feature=None)
b2 = SimpleNamespace(z=torch.tensor([1, 6, 8], device=DEV), pos=torch.tensor([[0., 0, 0], [50., 0, 0], [100., 0, 0]], device=DEV),
                     batch=torch.tensor([0, 1, 2], device=DEV), y=torch.zeros(3, device=DEV), node_feature=None)   # no edge at all
BATCHES = [('one pair', b), ('no edges', b2)]
for cls, kw in (('SphereNet', dict(hidden_channels=32, int_emb_size=16, out_emb_channels=32, num_spherical=3,
                                   num_radial=4, num_layers=2)),
                ('DimeNetPP', dict(hidden_channels=32, int_emb_size=16, out_emb_channels=32, num_spherical=3,
                                   num_radial=4, num_layers=2)),
                ('SchNet', dict(num_layers=2, hidden_channels=32, num_filters=32)),
                ('ComENet', dict(num_layers=2, hidden_channels=64, middle_channels=32))):
    for tag, bb in BATCHES:
        torch.manual_seed(0)
        m = getattr(M, cls)(**kw).to(DEV)
        try:
            out = m(bb)
            print(cls, tag, 'out finite', bool(torch.isfinite(out).all()), tuple(out.shape))
            out.sum().backward()
            for n, p in m.named_parameters():
                if p.grad is not None and not torch.isfinite(p.grad).all():
                    print('  NONFINITE', n, tuple(p.shape), int((~torch.isfinite(p.grad)).sum()))
        except Exception as e:
            print(cls, tag, 'RAISED', type(e).__name__, str(e)[:200])
