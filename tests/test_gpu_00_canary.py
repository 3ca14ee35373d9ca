"""GPU canary — collected FIRST (tests/conftest.py): names the device and walks the eager small-shape path one stage at a
time with a device synchronisation after each, so a box that faults says WHERE (the r03 driver run died 2.9 s into the
suite with no stage information; DESIGN.md "r03 driver fault")."""
import json
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mark(msg):
    sys.stderr.write(f'[canary] {msg}\n')
    sys.stderr.flush()


def test_canary_device_and_eager_path_stage_by_stage():
    from dig_amd import _hip
    from dig_amd.graph import start_graph
    from dig_amd.synthetic import make_batch, batch_to
    import dig_amd.threedgraph.method as M
    p = torch.cuda.get_device_properties(0)
    info = _hip.device_info()
    _mark('device ' + json.dumps(dict(name=p.name, gcn=getattr(p, 'gcnArchName', '?'), cus=p.multi_processor_count,
                                      mem_gib=round(p.total_memory / 2 ** 30, 1), lib=info,
                                      torch=torch.__version__, hip=torch.version.hip)))
    assert info['cus'] == p.multi_processor_count and info['wave'] == 64
    assert 'gfx950' in getattr(p, 'gcnArchName', 'gfx950'), p
    torch.manual_seed(0)
    model = M.SphereNet(hidden_channels=64, int_emb_size=32, out_emb_channels=64, num_spherical=3, num_radial=4,
                        num_layers=2).to('cuda:0')
    torch.cuda.synchronize()
    _mark('stage 1: model on device')
    b = batch_to(make_batch(4, 6, 12, 0.08, 5.0, seed=5), 'cuda:0')
    torch.cuda.synchronize()
    _mark('stage 2: batch on device')
    pend = start_graph(b.pos, b.batch, 5.0)
    torch.cuda.synchronize()
    _mark('stage 3: radius graph + CSR + triplet counts enqueued and complete')
    g = pend.finish()
    torch.cuda.synchronize()
    _mark(f'stage 4: triplet lists filled (N={g.N} E={g.E} T={g.T})')
    assert g.B == 4 and g.E > 0 and g.T > 0
    assert int(g.src.max()) < g.N and int(g.kj.max()) < g.E and int(g.ji.max()) < g.E
    out = model(b)
    torch.cuda.synchronize()
    _mark('stage 5: forward')
    assert torch.isfinite(out).all()
    (out - b.y.unsqueeze(1)).abs().mean().backward()
    torch.cuda.synchronize()
    _mark('stage 6: backward')
    assert all(torch.isfinite(q.grad).all() for q in model.parameters() if q.grad is not None)
