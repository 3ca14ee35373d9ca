"""TEST INFRASTRUCTURE — generate tests/golden/*.npz by running the reference's python
sources VERBATIM (oracle/ref_loader.py) in the build container.

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden spherenet_tiny schnet_cfg1_b32

Every fixture records what the *reference itself* produced (float32, the precision the
reference runs in; float64 additionally for the tiny cases) on the deterministic weights and
batches of tests/fixture_utils.py.  Loss follows run.py:126-131 (L1; + p*L1(force), p=100).
The GPU box has no /root/reference, so these files are the reference's voice there.
"""
import copy
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch, grad_sample_index   # noqa: E402
from oracle import ref_loader                                            # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def _run(model, batch, energy_and_force, dtype, full):
    from torch.autograd import grad
    m = copy.deepcopy(model).to(dtype)
    b = copy.copy(batch)
    if hasattr(batch, 'pos'):
        b.pos = batch.pos.to(dtype).clone()
        if torch.is_tensor(getattr(batch, 'node_feature', None)):
            b.node_feature = batch.node_feature.to(dtype)
    else:                                           # protein batch (ProNet): every float field in the run's precision
        for k in ('coords_ca', 'coords_n', 'coords_c', 'bb_embs', 'side_chain_embs'):
            setattr(b, k, getattr(batch, k).to(dtype).clone())
    if dtype != torch.float32 and hasattr(m, 'pos_emb'):
        # ProNet builds its positional embedding in hard-coded float32 (pronet.py:366-371); the float64 yardstick run
        # casts that one tensor, nothing else of the reference is touched
        orig = m.pos_emb
        m.pos_emb = lambda edge_index, num_pos_emb=16: orig(edge_index, num_pos_emb).to(dtype)
    m.zero_grad()
    out = m(b)
    y = batch.y.to(dtype)
    res = {}
    if energy_and_force:
        force = -grad(outputs=out, inputs=b.pos, grad_outputs=torch.ones_like(out),
                      create_graph=True, retain_graph=True)[0]
        loss = (out - y.unsqueeze(1)).abs().mean() + 100 * (force - batch.force.to(dtype)).abs().mean()
        res['force'] = force.detach().numpy()
    else:
        loss = (out - y.unsqueeze(1)).abs().mean()
    loss.backward()
    res['out'] = out.detach().numpy()
    res['loss'] = np.asarray(loss.item())
    for n, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        res['gnorm/' + n] = np.asarray(g.norm().item())
        res['gsamp/' + n] = g.detach().reshape(-1)[grad_sample_index(g.numel())].numpy()
        if full and g.numel() <= 2048:
            res['grad/' + n] = g.detach().numpy()
    return res


def make_case(name):
    warnings.filterwarnings('ignore')
    ref_loader.load()
    import digref.threedgraph.method as M
    from digref.threedgraph.utils import xyz_to_dat
    from oracle import pyg_shim as S
    cls, kw, bname, wseed = MODEL_CASES[name]
    t0 = time.time()
    torch.manual_seed(0)
    model = getattr(M, cls)(**kw)
    sd = det_state_dict(model.state_dict(), wseed)
    model.load_state_dict(sd)
    batch = get_batch(bname)
    eaf = bool(kw.get('energy_and_force', False))
    out = {'meta/case': np.asarray(name), 'meta/num_params': np.asarray(sum(p.numel() for p in model.parameters()))}
    small = 'tiny' in name
    r32 = _run(model, batch, eaf, torch.float32, small)
    out.update({'f32/' + k: v for k, v in r32.items()})
    if True:
        r64 = _run(model, batch, eaf, torch.float64, small)
        out.update({'f64/' + k: v for k, v in r64.items() if k in ('out', 'loss', 'force') or k.startswith('gnorm/') or k.startswith('gsamp/') or small})
    # graph + geometry intermediates, as the reference computes them (float32)
    cutoff = getattr(model, 'cutoff')
    pos = batch.pos if hasattr(batch, 'pos') else batch.coords_ca
    ei = S.radius_graph(pos, cutoff, batch.batch, max_num_neighbors=getattr(model, 'max_num_neighbors', 32))
    out['geom/E'] = np.asarray(ei.size(1))
    if cls in ('SphereNet', 'DimeNetPP'):
        tors = cls == 'SphereNet'
        r = xyz_to_dat(pos, ei, pos.size(0), use_torsion=tors)
        if tors:
            dist, angle, torsion, i, j, kj, ji = r
            out['geom/torsion_sum'] = np.asarray(torsion.double().sum().item())
        else:
            dist, angle, i, j, kj, ji = r
        out['geom/T'] = np.asarray(kj.numel())
        out['geom/idx_kj_sum'] = np.asarray(int((kj * (torch.arange(kj.numel()) % 1000 + 1)).sum()))
        out['geom/idx_ji_sum'] = np.asarray(int((ji * (torch.arange(ji.numel()) % 1000 + 1)).sum()))
        out['geom/edge_sum'] = np.asarray(int((ei[0] * 3 + ei[1] * 7).sum()))
        if small:
            out['geom/edge_index'] = ei.numpy()
            out['geom/dist'] = dist.numpy()
            out['geom/angle'] = angle.numpy()
            out['geom/idx_kj'] = kj.numpy()
            out['geom/idx_ji'] = ji.numpy()
            if tors:
                out['geom/torsion'] = torsion.numpy()
            with torch.no_grad():
                if tors:
                    e = model.emb(dist, angle, torsion, kj)
                    out['emb/rbf'], out['emb/sbf'], out['emb/tbf'] = (t.numpy() for t in e)
                else:
                    e = model.emb(dist, angle, kj)
                    out['emb/rbf'], out['emb/sbf'] = (t.numpy() for t in e)
    else:
        out['geom/edge_sum'] = np.asarray(int((ei[0] * 3 + ei[1] * 7).sum()))
        if small or cls in ('ComENet', 'ProNet'):
            out['geom/edge_index'] = ei.numpy().astype(np.int32)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(f'{name}: {time.time() - t0:.1f}s  out[:3]={r32["out"].ravel()[:3]}  loss={r32["loss"]}')


def make_ops():
    """Known answers recorded in the reference tree itself (not generated): kept here so the
    fixture directory is self-describing."""
    nb = dict(
        # examples/threedgraph/xyz_to_dat.ipynb cells 1-5
        edge_index=np.array([[1, 0, 2, 1, 3, 2], [0, 1, 1, 2, 2, 3]]),
        adj_row=np.array([0, 1, 1, 2, 2, 3]), adj_col=np.array([1, 0, 2, 1, 3, 2]),
        sel_row=np.array([0, 0, 1, 2, 2, 3, 3, 4, 5, 5]), sel_col=np.array([0, 2, 1, 1, 3, 0, 2, 2, 1, 3]),
        sel_val=np.array([1, 2, 0, 3, 4, 1, 2, 5, 3, 4]), num_triplets=np.array([2, 1, 2, 2, 1, 2]),
        idx_kj=np.array([2, 4, 1, 3]), idx_ji=np.array([0, 2, 3, 5]),
    )
    np.savez_compressed(os.path.join(GOLD, 'notebook_xyz_to_dat.npz'), **nb)


GSPHERE_KW = dict(cutoff=5.0, num_node_types=10, num_layers=2, hidden_channels=32, int_emb_size=16, basis_emb_size=4,
                  out_emb_channels=32, num_spherical=3, num_radial=4)
GSPHERE_CASES = {'gspherenet_tiny4': ('tiny4', 131), 'gspherenet_qm9_b8': ('qm9_b8', 132)}


def make_gspherenet(name):
    """G-SphereNet's private SphereNet (dig/ggraph3D/method/G_SphereNet/model/spherenet.py) run
    verbatim: node embeddings of forward / dist_only_forward, loss = mean |out|, gradient samples; float32 + float64."""
    warnings.filterwarnings('ignore')
    mod = ref_loader.load_gspherenet()
    bname, wseed = GSPHERE_CASES[name]
    torch.manual_seed(0)
    with torch.no_grad():       # features.py:181 writes into a Parameter with out= (rejected under autograd by torch 2.x)
        model = mod.SphereNet(**GSPHERE_KW)
    model.emb.dist_emb.freq.requires_grad_(True)      # ... and under no_grad that out= call clears the flag: restore it
    model.load_state_dict(det_state_dict(model.state_dict(), wseed))
    batch = get_batch(bname)
    out = {'meta/case': np.asarray(name)}
    for tag, dtype in (('f32', torch.float32), ('f64', torch.float64)):
        m = copy.deepcopy(model).to(dtype)
        m.zero_grad()
        o = m(batch.z, batch.pos.to(dtype), batch.batch)
        loss = o.abs().mean()
        loss.backward()
        out[tag + '/out'] = o.detach().numpy()
        out[tag + '/loss'] = np.asarray(loss.item())
        with torch.no_grad():
            out[tag + '/dist_only'] = m.dist_only_forward(batch.z, batch.pos.to(dtype), batch.batch).numpy()
        for n, p in m.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            out[f'{tag}/gsamp/{n}'] = g.detach().reshape(-1)[grad_sample_index(g.numel())].numpy()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(f'{name}: out[0,:3]={out["f32/out"][0, :3]} loss={out["f32/loss"]}')


if __name__ == '__main__':
    names = sys.argv[1:] or (list(MODEL_CASES) + list(GSPHERE_CASES))
    make_ops()
    for n in names:
        make_gspherenet(n) if n in GSPHERE_CASES else make_case(n)
