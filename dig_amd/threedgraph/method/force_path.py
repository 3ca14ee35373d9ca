"""Differentiable geometry + basis for ``energy_and_force=True`` (run.py:126: force = -d out / d pos with
``create_graph=True``, then ``loss.backward()`` differentiates THROUGH that gradient).

Everything between ``pos`` and the embeddings must therefore be twice differentiable.  The graph (integer
work) still comes from the HIP builder, and positions reach edges / triplets through the HIP row-gather /
segment-sum pair (``ops.gather_rows`` — its backward is a segment sum whose backward is the gather again, so
any order of differentiation stays on the engine's kernels).  The elementwise geometry and basis formulas
are written with torch ops here so autograd can differentiate them twice; analytic second-order HIP kernels
are the planned replacement (DESIGN.md "next").  Values agree with the fused forward kernels to float32
round-off (tests/test_gpu_models.py::test_force_path_matches_fused).
"""
import math

import torch

from ... import ops
from ...graph import csr_by_key


def _cross(a, b):
    return torch.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                        a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                        a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], dim=1)


def _bessel(dist, cutoff, ns, nr, zeros, norms, env_p):
    """[E, ns*nr] = norm * j_l(z * d/c) (* envelope); float64 internally like csrc/basis.hip."""
    x = (dist / cutoff).double().unsqueeze(1)                    # [E,1]
    u = zeros.view(1, -1) * x                                    # [E, ns*nr]
    s, c = torch.sin(u), torch.cos(u)
    jl = [s / u, s / (u * u) - c / u]
    for l in range(1, ns - 1):
        jl.append((2 * l + 1) / u * jl[l] - jl[l - 1])
    # column block l (width nr) takes j_l
    out = torch.cat([jl[l][:, l * nr:(l + 1) * nr] for l in range(ns)], 1) * norms.view(1, -1)
    if env_p > 0:
        p = env_p
        a, b, cc = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
        x0 = x.pow(p - 1)
        x1 = x0 * x
        out = out * (1.0 / x + a * x0 + b * x1 + cc * x1 * x)
    return out.float()


def _harmonics(theta, phi, ns, pref):
    """[M, ns] (phi None) or [M, ns*ns] real harmonics, same recurrences/order as csrc/basis.hip."""
    NSM = 8
    ct, st = torch.cos(theta), torch.sin(theta)
    P = [[None] * ns for _ in range(ns)]
    for m in range(ns):
        P[m][m] = torch.ones_like(ct) if m == 0 else (1 - 2 * m) * P[m - 1][m - 1]
        if m + 1 < ns:
            P[m + 1][m] = (2 * m + 1) * ct * P[m][m]
        for l in range(m + 2, ns):
            P[l][m] = ((2 * l - 1) * ct * P[l - 1][m] - (l + m - 1) * P[l - 2][m]) / (l - m)
        if phi is None:
            break
    if phi is None:
        return torch.stack([pref[l * NSM] * P[l][0] for l in range(ns)], 1)
    x, y = st * torch.cos(phi), st * torch.sin(phi)
    C, S_ = [torch.ones_like(x)], [torch.zeros_like(x)]
    for m in range(1, ns):
        S_.append(x * S_[m - 1] + y * C[m - 1])
        C.append(x * C[m - 1] - y * S_[m - 1])
    cols = [None] * (ns * ns)
    for l in range(ns):
        cols[l * l] = pref[l * NSM] * P[l][0]
        for m in range(1, l + 1):
            k = pref[l * NSM + m] * P[l][m]
            cols[l * l + m] = k * C[m]
            cols[l * l + 2 * l + 1 - m] = k * S_[m]
    return torch.stack(cols, 1)


def dime_geometry_differentiable(model, pos, g):
    """(rbf, sbf[, tbf]) for SphereNet / DimeNet++ as differentiable functions of ``pos``."""
    emb = model.emb
    ns, nr = emb.ns, emb.nr
    zeros, norms, pref = emb.tables.on(pos.device)
    vec = ops.gather_rows(pos, g.seg_dst) - ops.gather_rows(pos, g.seg_src)       # pos_i - pos_j  [E,3]
    if g.cnt_E is not None:
        # static-shape (HIP-graph) batch: the masked gathers give zero vectors in the padded rows, where sqrt / 1/x /
        # atan2 are singular.  Padded edges become the unit vector e_x (dist = 1), padded triplets the right angle
        # (e_x, e_y): every padded value and derivative is finite, every gradient entering a padded row is exactly 0.
        # (built from device-side ops only: a host-initialised tensor would be a copy inside the capture)
        pad_e = (torch.arange(g.E, device=pos.device) >= g.cnt_E).to(pos.dtype).unsqueeze(1)
        pad_t = (torch.arange(g.T, device=pos.device) >= g.cnt_T).to(pos.dtype).unsqueeze(1)
        ze, zt = torch.zeros_like(pad_e), torch.zeros_like(pad_t)
        vec = vec + torch.cat([pad_e, ze, ze], 1)
    dist = vec.pow(2).sum(-1).sqrt()
    v_ji = ops.gather_rows(vec, g.seg_ji)                                         # [T,3]
    v_jk = -ops.gather_rows(vec, g.seg_kj)                                        # pos_k - pos_j
    if g.cnt_E is not None:
        v_ji = v_ji + torch.cat([pad_t, zt, zt], 1)
        v_jk = v_jk + torch.cat([zt, pad_t, zt], 1)
    a = (v_ji * v_jk).sum(-1)
    b = _cross(v_ji, v_jk).norm(dim=-1)
    angle = torch.atan2(b, a)
    rbf = emb.dist_emb(dist)
    bes = _bessel(dist, emb.cutoff, ns, nr, zeros, norms, emb.env_p)              # [E, ns*nr]
    bes_t = ops.gather_rows(bes, g.seg_kj)                                        # rbf[idx_kj]
    sbf = (bes_t.view(-1, ns, nr) * _harmonics(angle, None, ns, pref).unsqueeze(-1)).reshape(-1, ns * nr)
    if not emb.torsion:
        return rbf, sbf
    # torsion = min over reference neighbours: the HIP kernel supplies the arg-min edge, the value is
    # recomputed differentiably for that neighbour only (gradient reaches the arg-min element, as in
    # torch_scatter's scatter(reduce='min') backward).
    with torch.no_grad():
        _, tor_k, targ = ops.triplet_geom(pos.detach().contiguous(), g, True)
    seg_n = csr_by_key(targ, g.E)
    v_jn = -ops.gather_rows(vec, seg_n)
    d_ji = v_ji.pow(2).sum(-1).sqrt()
    p1, p2 = _cross(v_ji, v_jk), _cross(v_ji, v_jn)
    ta = (p1 * p2).sum(-1)
    tb = (_cross(p1, p2) * v_ji).sum(-1) / d_ji
    tor = torch.atan2(tb, ta)
    tor = torch.where(tor <= 0, tor + 2 * math.pi, tor)
    # arg-min == the triplet's own k: the value is a float32 rounding residue of the reference's
    # arithmetic, analytically constant -> take the kernel's value, no gradient.
    tor = torch.where(targ == g.kj, tor_k, tor)
    ylm = _harmonics(angle, tor, ns, pref)
    tbf = (bes_t.view(-1, 1, ns, nr) * ylm.view(-1, ns, ns, 1)).reshape(-1, ns * ns * nr)
    return rbf, sbf, tbf
