"""G-SphereNet's private SphereNet (dig/ggraph3D/method/G_SphereNet/model/spherenet.py) on the engine
against what the verbatim reference produced (tests/golden/gspherenet_*.npz, oracle/make_golden.py:make_gspherenet):
node embeddings of ``forward`` and ``dist_only_forward``, loss, and every parameter's gradient samples."""
import os

import numpy as np
import pytest
import torch

from tests.fixture_utils import det_state_dict, get_batch, grad_sample_index

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
KW = dict(cutoff=5.0, num_node_types=10, num_layers=2, hidden_channels=32, int_emb_size=16, basis_emb_size=4,
          out_emb_channels=32, num_spherical=3, num_radial=4)
CASES = {'gspherenet_tiny4': ('tiny4', 131), 'gspherenet_qm9_b8': ('qm9_b8', 132)}


def test_state_dict_matches_the_reference_module():
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    import dig_amd.ggraph3D as G
    mod = ref_loader.load_gspherenet()
    with torch.no_grad():
        ref = mod.SphereNet(**KW)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in G.SphereNet(**KW).state_dict().items()}
    assert a == b, set(a) ^ set(b)
    import dig.ggraph3D.method.G_SphereNet.model.spherenet as alias        # the reference's import path
    assert alias.SphereNet is G.SphereNet


@pytest.mark.gpu
@pytest.mark.parametrize('case', list(CASES))
def test_gspherenet_matches_verbatim_reference(case):
    import dig_amd.ggraph3D as G
    from dig_amd.synthetic import batch_to
    bname, wseed = CASES[case]
    gold = np.load(os.path.join(GOLD, case + '.npz'))
    model = G.SphereNet(**KW)
    model.load_state_dict(det_state_dict(model.state_dict(), wseed))
    model = model.cuda()
    b = batch_to(get_batch(bname), 'cuda')
    out = model(b.z, b.pos, b.batch)
    loss = out.abs().mean()
    loss.backward()
    scale = np.abs(gold['f64/out']).max()
    noise = np.abs(gold['f32/out'] - gold['f64/out']).max() / scale
    err = np.abs(out.detach().cpu().numpy() - gold['f64/out']).max() / scale
    assert err <= max(1e-5, 3 * noise), (err, noise)
    assert abs(loss.item() - float(gold['f64/loss'])) <= 1e-5 * abs(float(gold['f64/loss']))
    with torch.no_grad():
        d = model.dist_only_forward(b.z, b.pos, b.batch).cpu().numpy()
    assert np.abs(d - gold['f64/dist_only']).max() <= 1e-5 * np.abs(gold['f64/dist_only']).max()
    names = [n for n, _ in model.named_parameters()]
    gm = max(np.abs(gold['f64/gsamp/' + n]).max() for n in names)
    worst = gnoise = 0.0
    bad = {}
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        mine = g.reshape(-1)[grad_sample_index(g.numel())].cpu().numpy()
        e = np.abs(mine - gold['f64/gsamp/' + n]).max() / gm
        if e > 1e-5:
            bad[n] = float(e)
        worst = max(worst, e)
        gnoise = max(gnoise, np.abs(gold['f32/gsamp/' + n] - gold['f64/gsamp/' + n]).max() / gm)
    assert worst <= max(1e-5, 3 * gnoise), (worst, gnoise, bad)
