"""INTEGRATION.md §2 — the operator-level drop-in: the reference's model code keeps running and only the three
third-party call sites are rebound to the engine's PUBLIC operators

    torch_geometric.nn.radius_graph  ->  dig_amd.ops.radius_graph          (spherenet.py:304)
    utils.xyz_to_dat                 ->  dig_amd.threedgraph.utils.xyz_to_dat   (spherenet.py:305, differentiable: :302 +
                                                                                  run.py:126 train forces through it)
    torch_scatter.scatter            ->  dig_amd.ops.scatter                (spherenet.py:171,211,224)

The GPU box has no /root/reference, so the network here is the restated oracle (float32, plain torch ops on the GPU)
with exactly those three names rebound; energies AND forces must reproduce what the verbatim reference recorded
(tests/golden: float32 + float64 runs)."""
import os

import numpy as np
import pytest
import torch

from oracle import threedgraph_oracle as O
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
from tests.test_oracle_golden import oracle_kwargs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('case', ['spherenet_force_md17_b8', 'dimenetpp_force_md17_b8', 'spherenet_tiny'])
def test_reference_network_on_engine_operators(case):
    import dig_amd.threedgraph.method as M
    from dig_amd import ops
    from dig_amd.synthetic import batch_to
    from dig_amd.threedgraph.utils import xyz_to_dat
    cls, kw, bname, wseed = MODEL_CASES[case]
    gold = np.load(os.path.join(GOLD, case + '.npz'))
    eaf = bool(kw.get('energy_and_force', False))
    sd = det_state_dict(getattr(M, cls)(**kw).state_dict(), wseed)
    sd = {k: v.cuda() for k, v in sd.items()}
    b = batch_to(get_batch(bname), 'cuda')
    pos = b.pos.clone().requires_grad_(eaf)
    cutoff = kw.get('cutoff', 5.0)
    edge_index = ops.radius_graph(pos.detach(), cutoff, b.batch)                  # engine op 1

    def scatter_fn(src, index, dim=0, dim_size=None):                             # engine op 3
        return ops.scatter(src, index, dim=dim, dim_size=dim_size, reduce='sum')

    fwd = O.spherenet_forward if cls == 'SphereNet' else O.dimenetpp_forward
    out = fwd(sd, b.z, pos, b.batch, dtype=torch.float32, geom_dtype=torch.float32, edge_index=edge_index,
              geom_fn=xyz_to_dat, scatter_fn=scatter_fn, **oracle_kwargs(cls, kw))       # engine op 2 = geom_fn
    scale = np.abs(gold['f64/out']).max()
    noise = np.abs(gold['f32/out'] - gold['f64/out']).max() / scale
    err = np.abs(out.detach().cpu().numpy() - gold['f64/out']).max() / scale
    assert err <= max(1e-5, 3 * noise), (err, noise)
    if eaf:
        force = -torch.autograd.grad(out, pos, torch.ones_like(out), create_graph=True)[0]
        fs = np.abs(gold['f64/force']).max()
        fnoise = np.abs(gold['f32/force'] - gold['f64/force']).max() / fs
        ferr = np.abs(force.detach().cpu().numpy() - gold['f64/force']).max() / fs
        assert ferr <= max(1e-5, 3 * fnoise), (ferr, fnoise)
        # ... and the loss of run.py:126-131 differentiates through that force (second order through xyz_to_dat)
        loss = (out - b.y.unsqueeze(1)).abs().mean() + 100 * (force - b.force).abs().mean()
        assert abs(loss.item() - float(gold['f32/loss'])) <= 2e-4 * abs(float(gold['f32/loss']))
        (gp,) = torch.autograd.grad(loss, pos)
        assert torch.isfinite(gp).all() and gp.abs().max() > 0
