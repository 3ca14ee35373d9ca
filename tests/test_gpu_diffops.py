"""GPU: the twice-differentiable operator set (dig_amd/diffops.py, csrc/diffgeom.hip) against float64 torch autograd
of the same formulas — values, first derivatives and second derivatives (gradient of a gradient contraction), i.e.
exactly what ``force = -grad(E, pos, create_graph=True); loss.backward()`` (run.py:126-133) exercises."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _second_order(fn_hip, fn_ref, inputs, seed=0):
    """inputs: list of float32 cuda tensors (requires_grad set by us).  Compares value, d(sum w*f)/dx and
    d/dx,w' of <d(sum w*f)/dx, v> between the HIP Function (float32) and the float64 reference."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    xs32 = [x.detach().clone().requires_grad_() for x in inputs]
    xs64 = [x.detach().double().clone().requires_grad_() for x in inputs]
    o32 = fn_hip(*xs32)
    o64 = fn_ref(*xs64)
    o32 = o32 if isinstance(o32, (tuple, list)) else (o32,)
    o64 = o64 if isinstance(o64, (tuple, list)) else (o64,)
    res = {}
    ws = [torch.randn(o.shape, generator=g).to(DEV) for o in o64]
    vs = [torch.randn(x.shape, generator=g).to(DEV) for x in xs64]
    for k, (a, b) in enumerate(zip(o32, o64)):
        res[f'val{k}'] = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    L32 = sum((o * w.float()).sum() for o, w in zip(o32, ws))
    L64 = sum((o * w).sum() for o, w in zip(o64, ws))
    g32 = torch.autograd.grad(L32, xs32, create_graph=True)
    g64 = torch.autograd.grad(L64, xs64, create_graph=True)
    for k, (a, b) in enumerate(zip(g32, g64)):
        res[f'grad{k}'] = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    # second order: differentiate <grad, v> plus a term through the outputs' weights
    M32 = sum((a * v.float()).sum() for a, v in zip(g32, vs))
    M64 = sum((a * v).sum() for a, v in zip(g64, vs))
    h32 = torch.autograd.grad(M32, xs32, allow_unused=True)
    h64 = torch.autograd.grad(M64, xs64, allow_unused=True)
    for k, (a, b) in enumerate(zip(h32, h64)):
        if b is None:
            continue
        a = a if a is not None else torch.zeros_like(xs32[k])
        res[f'hess{k}'] = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    return res


def _small_graph(tors=True, seed=3):
    from dig_amd.graph import build_graph
    from dig_amd.synthetic import make_batch, batch_to
    b = batch_to(make_batch(3, 6, 10, 0.08, 5.0, seed=seed), DEV)
    return b, build_graph(b.pos, b.batch, 5.0, triplets=True)


def _cross(a, b):
    return torch.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                        a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)


@pytest.mark.parametrize('tors', [False, True])
def test_geometry_first_and_second_derivatives(tors):
    from dig_amd import diffops, ops
    b, g = _small_graph()
    src, dst, kj, ji = (t.long() for t in (g.src, g.dst, g.kj, g.ji))
    with torch.no_grad():
        _, tor_val, targ = ops.triplet_geom(b.pos.contiguous(), g, True)
    targ = targ.long()
    live = (targ >= 0) & (targ != kj)

    def hip(pos):
        vec = diffops.edge_vectors(pos, g)
        d = diffops.edge_len(vec, 0, None)
        if tors:
            a, t = diffops.triplet_angles(vec, pos.detach().contiguous(), g, True)
            return d, a, t
        return d, diffops.triplet_angles(vec, pos.detach().contiguous(), g, False)

    def ref(pos):
        vec = pos[dst] - pos[src]
        d = vec.pow(2).sum(-1).sqrt()
        v1, v2 = vec[ji], -vec[kj]
        a = torch.atan2(_cross(v1, v2).norm(dim=-1), (v1 * v2).sum(-1))
        if not tors:
            return d, a
        v3 = -vec[targ.clamp(min=0)]
        p1, p2 = _cross(v1, v2), _cross(v1, v3)
        ta = (p1 * p2).sum(-1)
        tb = (_cross(p1, p2) * v1).sum(-1) / v1.pow(2).sum(-1).sqrt()
        t = torch.atan2(tb, ta)
        t = torch.where(t <= 0, t + 2 * math.pi, t)
        t = torch.where(live, t, tor_val.double())            # residue / missing cases: the kernel's constant
        return d, a, t

    res = _second_order(hip, ref, [b.pos])
    assert res['val0'] < 1e-6 and res['val1'] < 2e-6, res
    if tors:
        assert res['val2'] < 1e-5, res
    assert res['grad0'] < 2e-5 and res['hess0'] < 2e-5, res


def test_public_xyz_to_dat_is_differentiable_on_unsorted_edges():
    """the reference model calls pos.requires_grad_() and differentiates through xyz_to_dat; an
    arbitrary (unsorted) edge_index takes the CSR-position -> edge-id map of the torsion arg-min."""
    from dig_amd import ops
    from dig_amd.threedgraph.utils import xyz_to_dat
    b, g = _small_graph(seed=5)
    ei = g.edge_index
    perm = torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(1)).to(DEV)
    ei = ei[:, perm].contiguous()
    N = b.pos.size(0)
    pos = b.pos.clone().requires_grad_()
    dist, angle, torsion, i, j, kj, ji = xyz_to_dat(pos, ei, N, use_torsion=True)
    p64 = b.pos.double().requires_grad_()
    d64 = (p64[i] - p64[j]).pow(2).sum(-1).sqrt()
    ti, tj, tk = i[ji], j[ji], j[kj]
    v1, v2 = p64[ti] - p64[tj], p64[tk] - p64[tj]
    a64 = torch.atan2(_cross(v1, v2).norm(dim=-1), (v1 * v2).sum(-1))
    assert (dist.double() - d64).abs().max() < 1e-6 and (angle.double() - a64).abs().max() < 1e-5
    w = torch.randn(dist.numel(), generator=torch.Generator().manual_seed(2)).to(DEV)
    u = torch.randn(angle.numel(), generator=torch.Generator().manual_seed(3)).to(DEV)
    (g32,) = torch.autograd.grad((dist * w).sum() + (angle * u).sum() + torsion.sum() * 0, pos, retain_graph=True)
    (g64,) = torch.autograd.grad((d64 * w.double()).sum() + (a64 * u.double()).sum(), p64)
    assert (g32.double() - g64).abs().max().item() <= 2e-5 * g64.abs().max().item()
    (gt,) = torch.autograd.grad(torsion.sum(), pos)            # runs, finite (values checked in test_gpu_ops)
    assert torch.isfinite(gt).all()
    # forward values are the forward-only kernels' values, bit for bit
    d0, a0, t0 = xyz_to_dat(b.pos, ei, N, use_torsion=True)[:3]
    assert torch.equal(d0, dist.detach()) and torch.equal(a0, angle.detach()) and torch.equal(t0, torsion.detach())


@pytest.mark.parametrize('env_p', [0, 6])
def test_bessel_table_derivatives(env_p):
    from dig_amd import diffops
    from dig_amd.threedgraph.method.basis import BasisTables
    ns, nr, cutoff = 7, 6, 5.0
    zeros, norms, pref = BasisTables(ns, nr, 'spherenet').on(torch.device(DEV))
    dist = (torch.rand(300, generator=torch.Generator().manual_seed(0)) * 3.9 + 0.9).to(DEV)

    def ref(d):
        x = (d / cutoff).unsqueeze(1)
        u = zeros.view(1, -1) * x
        s, c = torch.sin(u), torch.cos(u)
        jl = [s / u, s / (u * u) - c / u]
        for l in range(1, ns - 1):
            jl.append((2 * l + 1) / u * jl[l] - jl[l - 1])
        out = torch.cat([jl[l][:, l * nr:(l + 1) * nr] for l in range(ns)], 1) * norms.view(1, -1)
        if env_p > 0:
            p = env_p
            a, b, cc = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
            x0 = x.pow(p - 1)
            out = out * (1.0 / x + a * x0 + b * x0 * x + cc * x0 * x * x)
        return out

    res = _second_order(lambda d: diffops.bessel_basis(d, cutoff, ns, nr, zeros, norms, env_p, None), ref, [dist])
    assert res['val0'] < 2e-6 and res['grad0'] < 1e-5 and res['hess0'] < 1e-5, res


@pytest.mark.parametrize('ns,with_phi', [(7, False), (7, True), (3, True), (2, False)])
def test_harmonics_derivatives(ns, with_phi):
    from dig_amd import diffops
    from dig_amd.threedgraph.method.basis import BasisTables
    pref = BasisTables(ns, 6, 'spherenet').on(torch.device(DEV))[2]
    gen = torch.Generator().manual_seed(1)
    theta = (torch.rand(257, generator=gen) * 2.6 + 0.2).to(DEV)
    phi = (torch.rand(257, generator=gen) * 6.0 + 0.1).to(DEV)
    NSM = 8

    def ref(th, ph=None):
        ct, st = torch.cos(th), torch.sin(th)
        P = [[None] * ns for _ in range(ns)]
        for m in range(ns):
            P[m][m] = torch.ones_like(ct) if m == 0 else (1 - 2 * m) * P[m - 1][m - 1]
            if m + 1 < ns:
                P[m + 1][m] = (2 * m + 1) * ct * P[m][m]
            for l in range(m + 2, ns):
                P[l][m] = ((2 * l - 1) * ct * P[l - 1][m] - (l + m - 1) * P[l - 2][m]) / (l - m)
            if ph is None:
                break
        if ph is None:
            return torch.stack([pref[l * NSM].double() * P[l][0] for l in range(ns)], 1)
        x, y = st * torch.cos(ph), st * torch.sin(ph)
        C, S_ = [torch.ones_like(x)], [torch.zeros_like(x)]
        for m in range(1, ns):
            S_.append(x * S_[m - 1] + y * C[m - 1])
            C.append(x * C[m - 1] - y * S_[m - 1])
        cols = [None] * (ns * ns)
        for l in range(ns):
            cols[l * l] = pref[l * NSM].double() * P[l][0]
            for m in range(1, l + 1):
                k = pref[l * NSM + m].double() * P[l][m]
                cols[l * l + m] = k * C[m]
                cols[l * l + 2 * l + 1 - m] = k * S_[m]
        return torch.stack(cols, 1)

    if with_phi:
        res = _second_order(lambda t, p: diffops.harmonics(t, p, ns, pref, None), ref, [theta, phi])
        assert max(res['grad1'], res['hess1']) < 2e-5, res
    else:
        res = _second_order(lambda t: diffops.harmonics(t, None, ns, pref, None), ref, [theta])
    assert res['val0'] < 5e-6 and res['grad0'] < 2e-5 and res['hess0'] < 2e-5, res


def test_dist_emb_derivatives_including_freq():
    from dig_amd import diffops
    cutoff, p, nr = 5.0, 6, 6
    gen = torch.Generator().manual_seed(2)
    dist = (torch.rand(700, generator=gen) * 3.9 + 0.9).to(DEV)
    freq = (torch.arange(1, nr + 1).float() * math.pi + 0.01 * torch.randn(nr, generator=gen)).to(DEV)

    def ref(d, f):
        a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
        x = d.unsqueeze(-1) / cutoff
        x0 = x.pow(p - 1)
        x1 = x0 * x
        env = 1.0 / x + a * x0 + b * x1 + c * (x1 * x)
        return env * (f * x).sin()

    res = _second_order(lambda d, f: diffops.dist_emb(d, f, cutoff, p, None), ref, [dist, freq])
    assert res['val0'] < 2e-6, res
    assert max(res['grad0'], res['grad1'], res['hess0'], res['hess1']) < 2e-5, res


@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('shape', [(700, 128, 128), (333, 6, 128), (257, 64, 256)])
def test_dense_layer_second_order(act, shape):
    from dig_amd import diffops
    M, K, N = shape
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(M, K, generator=gen).to(DEV)
    W = (torch.randn(N, K, generator=gen) / math.sqrt(K)).to(DEV)
    bias = (0.1 * torch.randn(N, generator=gen)).to(DEV)
    r = torch.randn(M, N, generator=gen).to(DEV)

    def ref(x_, W_, b_, r_):
        z = x_ @ W_.t() + b_
        y = z if act == 0 else (torch.nn.functional.silu(z) if act == 1 else torch.nn.functional.softplus(z) - math.log(2.0))
        return r_ + y

    res = _second_order(lambda x_, W_, b_, r_: diffops.linear2(x_, W_, b_, act, r_), ref, [x, W, bias, r])
    tol = 3e-5
    assert res['val0'] < 1e-5, res
    for k in ('grad0', 'grad1', 'grad2', 'grad3', 'hess0', 'hess1'):
        assert res[k] < tol, (k, res)
    if act != 0:
        assert res['hess2'] < tol, res


def test_gather_mul_segsum_family_second_order():
    from dig_amd import diffops
    b, g = _small_graph()
    C = 16
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(g.E, C, generator=gen).to(DEV)
    A = torch.randn(g.T, C, generator=gen).to(DEV)
    kj, ji = g.kj.long(), g.ji.long()

    def ref(X_, A_):
        m = X_[kj] * A_
        return torch.zeros(g.E, C, dtype=m.dtype, device=m.device).index_add(0, ji, m) ** 2     # nonlinearity on top

    res = _second_order(lambda X_, A_: diffops.gather_mul_segsum(X_, A_, g.seg_kj, g.seg_ji) ** 2, ref, [X, A])
    assert res['val0'] < 1e-5 and max(res['grad0'], res['grad1'], res['hess0'], res['hess1']) < 2e-5, res


def test_elementwise_product_second_order():
    """diffops.mul2 (three kernels: forward, backward, double backward) against float64 autograd of a * b: value, both
    first derivatives, and the second-order terms (d/da, d/db of <grad, v>) — what the x_kj * radial-projection and
    e2 = lin_rbf(rbf) * e1 products of the energy_and_force route need (run.py:126-133)."""
    from dig_amd import diffops
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(1001, 128, generator=gen).to(DEV)
    b = torch.randn(1001, 128, generator=gen).to(DEV)
    res = _second_order(lambda x, y: diffops.mul2(x, y), lambda x, y: x * y, [a, b])
    assert res['val0'] < 1e-6 and res['grad0'] < 1e-6 and res['grad1'] < 1e-6, res
    assert res['hess0'] < 1e-6 and res['hess1'] < 1e-6, res
    # shapes that do not match fall back to the framework product
    c = torch.randn(1001, 1, generator=gen).to(DEV)
    assert torch.equal(diffops.mul2(a, c), a * c)


def test_fan_out_sums_the_consumers_gradients_in_one_launch_to_second_order():
    """diffops.fan_out / dig3d_sum_many: n aliases of x, each through its own nonlinear consumer — value, gradient and the
    gradient of a gradient contraction equal float64 autograd of the same expression on x itself; the create_graph backward
    runs ONE k_sum_many launch where autograd would run n - 1 additions."""
    from dig_amd import diffops
    n = 5
    x = torch.randn(700, 6, generator=torch.Generator().manual_seed(3)).to(DEV)

    def hip(x):
        xs = diffops.fan_out(x, n)
        assert len(xs) == n and all(a.data_ptr() == x.data_ptr() for a in xs)
        return sum(((k + 1.0) * a).sin() * a for k, a in enumerate(xs))

    def ref(x):
        return sum(((k + 1.0) * x).sin() * x for k in range(n))

    res = _second_order(hip, ref, [x])
    assert max(res.values()) <= 5e-6, res
    # the sum itself, bit for bit in the documented order
    gs = [torch.randn(700, 6, generator=torch.Generator().manual_seed(10 + k)).to(DEV) for k in range(n)]
    s = diffops._SumMany.apply(*gs)
    want = gs[0]
    for g in gs[1:]:
        want = want + g
    assert torch.equal(s, want)
    # fewer than three consumers: nothing to gain, the tensor itself is handed out
    y = x.clone().requires_grad_()
    assert all(a is y for a in diffops.fan_out(y, 2))
