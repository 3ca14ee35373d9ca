"""Seeded synthetic molecule batches (SURVEY.md §8(d) "Synthetic inputs").

There is no network, hence no QM9 / MD17 / OC20 files: every benchmark and parity test runs
on random-coordinate molecules of the same shape.  Per molecule: ``n`` atoms uniform in
[n_min, n_max]; positions rejection-sampled uniformly in a cube of side (n/rho)^(1/3) A with
minimum pair distance >= ``min_dist`` and every atom having >= ``min_neighbors`` neighbours
inside ``cutoff``; z ~ U{1..9}; y ~ N(0,1); force ~ N(0,1).

Host-side numpy only (PCG64 stream => identical batches on every machine).
"""
from types import SimpleNamespace

import numpy as np
import torch


def _one_molecule(rng, n, rho, min_dist, cutoff, min_neighbors, max_tries=200):
    side = (n / rho) ** (1.0 / 3.0)
    md2 = min_dist * min_dist
    for _ in range(max_tries):
        pts = np.empty((n, 3), dtype=np.float64)
        k = 0
        fails = 0
        while k < n and fails < 10000:
            cand = rng.random((16, 3)) * side
            for c in cand:
                if k == n:
                    break
                if k == 0 or ((pts[:k] - c) ** 2).sum(1).min() >= md2:
                    pts[k] = c
                    k += 1
                else:
                    fails += 1
        if k < n:
            continue
        d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        nb = (d2 < cutoff * cutoff).sum(1) - 1
        if nb.min() >= min_neighbors:
            return pts
    raise RuntimeError('could not place a molecule; lower rho or min_dist')


def make_batch(num_graphs, n_min, n_max, rho, cutoff, seed, min_dist=0.9, min_neighbors=2,
               with_force=False, device='cpu', node_feature_dim=0):
    """Returns a batch object with the attributes DIG's models read (run.py:123-131):
    z int64 [N], pos f32 [N,3], batch int64 [N] (sorted), ptr int64 [B+1], y f32 [B],
    optionally force f32 [N,3] and node_feature f32 [N, node_feature_dim] (SphereNet's ``use_extra_node_feature``,
    spherenet.py:259-264); plus num_graphs."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos, z, bvec, ptr = [], [], [], [0]
    for g in range(num_graphs):
        n = int(rng.integers(n_min, n_max + 1))
        pos.append(_one_molecule(rng, n, rho, min_dist, cutoff, min_neighbors))
        z.append(rng.integers(1, 10, size=n))
        bvec.append(np.full(n, g, dtype=np.int64))
        ptr.append(ptr[-1] + n)
    pos = np.concatenate(pos).astype(np.float32)
    N = pos.shape[0]
    out = SimpleNamespace(
        z=torch.from_numpy(np.concatenate(z).astype(np.int64)).to(device),
        pos=torch.from_numpy(pos).to(device),
        batch=torch.from_numpy(np.concatenate(bvec)).to(device),
        ptr=torch.tensor(ptr, dtype=torch.int64, device=device),
        y=torch.from_numpy(rng.standard_normal(num_graphs).astype(np.float32)).to(device),
        num_graphs=num_graphs,
        node_feature=None,
        ptr_list=list(ptr),          # host copy of the graph pointer (splitting a batch without a device read)
    )
    if with_force:
        out.force = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32)).to(device)
    if node_feature_dim:      # drawn last: the other fields of a seed do not depend on it
        out.node_feature = torch.from_numpy(rng.standard_normal((N, node_feature_dim)).astype(np.float32)).to(device)
    return out


def batch_to(b, device):
    out = SimpleNamespace(**vars(b))
    for k, v in vars(b).items():
        if torch.is_tensor(v):
            setattr(out, k, v.to(device))
    return out


# The five BASELINE.json configurations as generator arguments.
WORKLOADS = {
    'qm9_like_b32':   dict(num_graphs=32, n_min=9, n_max=29, rho=0.08, seed=1),
    'md17_like_b32':  dict(num_graphs=32, n_min=21, n_max=21, rho=0.09, seed=2, with_force=True),
    'oc20_like_b32':  dict(num_graphs=32, n_min=40, n_max=120, rho=0.05, seed=3),
    'atoms128_b128':  dict(num_graphs=128, n_min=128, n_max=128, rho=0.05, seed=4),
}


def make_protein_batch(num_graphs, n_min, n_max, seed, device='cpu'):
    """Synthetic protein chains for ProNet (method/pronet/pronet.py:374-381 reads x, coords_ca, coords_n, coords_c,
    bb_embs, side_chain_embs, batch, y): CA atoms on a self-avoiding-ish random walk with 3.8 A steps (residues
    i-1, i, i+1 never collinear), N / C atoms 1.46 / 1.52 A from their CA in random non-parallel directions, residue
    types U{0..25}, backbone / side-chain torsion embeddings = sin / cos of random angles."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ca, cn, cc, aa, bvec, ptr = [], [], [], [], [], [0]
    for g in range(num_graphs):
        n = int(rng.integers(n_min, n_max + 1))
        p = np.zeros((n, 3))
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        for k in range(1, n):
            while True:
                t = d + 0.9 * rng.standard_normal(3)
                t /= np.linalg.norm(t)
                cand = p[k - 1] + 3.8 * t
                if k < 2 or np.min(np.linalg.norm(p[:k - 1] - cand, axis=1)) > 3.0:
                    break
            p[k], d = cand, t
        def around(r):
            v = rng.standard_normal((n, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            return p + r * v
        ca.append(p)
        cn.append(around(1.46))
        cc.append(around(1.52))
        aa.append(rng.integers(0, 26, size=(n, 1)))
        bvec.append(np.full(n, g, dtype=np.int64))
        ptr.append(ptr[-1] + n)
    N = ptr[-1]
    ang_bb = rng.uniform(-np.pi, np.pi, size=(N, 3))
    ang_sc = rng.uniform(-np.pi, np.pi, size=(N, 4))
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)
    return SimpleNamespace(
        x=torch.from_numpy(np.concatenate(aa).astype(np.int64)).to(device),
        coords_ca=f32(np.concatenate(ca)), coords_n=f32(np.concatenate(cn)), coords_c=f32(np.concatenate(cc)),
        bb_embs=f32(np.concatenate([np.sin(ang_bb), np.cos(ang_bb)], 1)),
        side_chain_embs=f32(np.concatenate([np.sin(ang_sc), np.cos(ang_sc)], 1)),
        batch=torch.from_numpy(np.concatenate(bvec)).to(device),
        ptr=torch.tensor(ptr, dtype=torch.int64, device=device),
        y=f32(rng.standard_normal(num_graphs)), num_graphs=num_graphs, ptr_list=list(ptr))
