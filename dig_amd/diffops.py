"""Twice-differentiable operator set of the engine — what ``energy_and_force=True`` needs.

The reference obtains forces as ``-grad(out, pos, create_graph=True)`` and then calls ``loss.backward()``
(method/run.py:126-133): every op between ``pos`` and the energy is differentiated twice.  Its autograd does that
over hundreds of small ATen kernels per quantity (the second-order graph of a silu alone is ~10 launches).  Here each
op is a ``torch.autograd.Function`` whose backward is itself a Function on HIP kernels, so a training step of
DimeNet++ with forces is ~10^3 launches instead of ~10^4:

    geometry     vec -> dist, angle, torsion          csrc/diffgeom.hip  (derivatives by forward-mode duals, float64)
    basis        dist -> Bessel table, (theta, phi) -> harmonics, dist_emb (learnable freq)
    dense        y = act(x W^T + b) (+ res)           csrc/dense.hip MFMA kernels + two elementwise kernels
    aggregation  F(X, A) = sum_seg X[gather] * A  and  P(G, X) = G[.] * X[.]   closed under differentiation

First-order users (energy-only training) go through dig_amd/ops.py's fused kernels; the dist_emb Function below serves
both routes.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip
from ._hip import call, ptr
from .graph import csr_by_key, _stream

ACT_NONE = 0
ACT_SWISH = 1


def _c(t):
    if t.dtype != torch.float32:
        raise RuntimeError(f'expected float32 tensor, got {t.dtype}')
    if not t.is_cuda:
        raise _hip.Dig3dError('dig_amd op received a CPU tensor; the engine has no CPU fallback')
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------------------
# elementwise product, closed under differentiation: three kernels per product and step (forward, backward, double
# backward) where the autograd of ``a * b`` under create_graph issues ~9 framework multiplies / additions
# ---------------------------------------------------------------------------------------------------------------
class _Mul2(Function):
    """(y, a', b') = (a b, alias of a, alias of b).  The aliases are what the create_graph backward differentiates through: the
    gradients that its OWN backward sends to a and b in the final pass then arrive here as arguments and are added inside
    k_ew_mul_bwd — not by a framework addition per operand and product (tools/diag_force_fanin.py: 8 of a DimeNet++ step's
    [E, 128] additions)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        call('dig3d_ew_mul', ptr(a), ptr(b), ptr(y), a.numel(), _stream())
        a2, b2 = a.view_as(a), b.view_as(b)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a, b, a2, b2)
        return y, a2, b2

    @staticmethod
    def backward(ctx, g, ga2, gb2):
        a, b, a2, b2 = ctx.saved_tensors
        if g is None and ga2 is None and gb2 is None:
            return None, None
        if torch.is_grad_enabled():                   # create_graph: differentiable, through the aliases
            if g is None:
                return ga2, gb2
            ga, gb = _Mul2Bwd.apply(g, a2, b2)
            return (ga if ga2 is None else ga + ga2), (gb if gb2 is None else gb + gb2)
        ga, gb = torch.empty_like(a), torch.empty_like(a)
        call('dig3d_ew_mul_bwd', ptr(_c(g)) if g is not None else None, ptr(a), ptr(b), ptr(ga), ptr(gb), a.numel(),
             ptr(_c(ga2)) if ga2 is not None else None, ptr(_c(gb2)) if gb2 is not None else None, _stream())
        return ga, gb


class _Mul2Bwd(Function):
    @staticmethod
    def forward(ctx, g, a, b):
        g = _c(g)
        ga, gb = torch.empty_like(a), torch.empty_like(a)
        call('dig3d_ew_mul_bwd', ptr(g), ptr(a), ptr(b), ptr(ga), ptr(gb), a.numel(), None, None, _stream())
        ctx.save_for_backward(g, a, b)
        return ga, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, gga, ggb):
        g, a, b = ctx.saved_tensors
        if gga is None and ggb is None:
            return None, None, None
        og, oa, ob = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        call('dig3d_ew_mul_bwd2', ptr(_c(gga)) if gga is not None else None, ptr(_c(ggb)) if ggb is not None else None,
             ptr(g), ptr(a), ptr(b), ptr(og), ptr(oa), ptr(ob), a.numel(), _stream())
        return og, oa, ob


def mul2(a, b):
    """a * b for same-shape float32 GPU tensors, twice differentiable on three kernels; anything else: ``a * b``."""
    if (torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.is_cuda and b.is_cuda
            and a.dtype == b.dtype == torch.float32 and a.numel() > 0):
        return _Mul2.apply(a, b)[0]
    return a * b


# ---------------------------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------------------------
def _combine(vec, gg, g_dist, E, tptr=None, gv1=None, seg2=None, gv2=None, seg3=None, gv3=None, cnt=None, want_gd=False):
    out = torch.empty(E, 3, dtype=torch.float32, device=vec.device)
    o_gd = torch.empty(E, dtype=torch.float32, device=vec.device) if want_gd else None
    call('dig3d_edge_combine', ptr(vec), ptr(gg), ptr(g_dist), E, ptr(tptr), ptr(gv1),
         ptr(seg2.kptr) if seg2 is not None else None, ptr(seg2.perm) if seg2 is not None else None, ptr(gv2),
         ptr(seg3.kptr) if seg3 is not None else None, ptr(seg3.perm) if seg3 is not None else None, ptr(gv3),
         ptr(out), ptr(o_gd), ptr(cnt), _stream())
    return out, o_gd


class _EdgeLen(Function):
    """dist[e] = |vec[e]| in the reference's float32 operation order (geometric_computing.py:25 / schnet.py:158)."""

    @staticmethod
    def forward(ctx, vec, mode, cnt):
        vec = _c(vec)
        E = vec.size(0)
        dist = torch.empty(E, dtype=torch.float32, device=vec.device)
        call('dig3d_vec_len', ptr(vec), E, int(mode), ptr(dist), ptr(cnt), 1.0, _stream())
        ctx.cnt = cnt
        ctx.save_for_backward(vec)
        return dist

    @staticmethod
    def backward(ctx, g):
        (vec,) = ctx.saved_tensors
        return _EdgeLenBwd.apply(vec, g, ctx.cnt), None, None


class _EdgeLenBwd(Function):
    @staticmethod
    def forward(ctx, vec, g, cnt):
        g = _c(g)
        ctx.cnt = cnt
        ctx.save_for_backward(vec, g)
        return _combine(vec, None, g, vec.size(0), cnt=cnt)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gg):
        vec, g = ctx.saved_tensors
        out, o_gd = _combine(vec, _c(gg), g, vec.size(0), cnt=ctx.cnt, want_gd=True)
        return out, o_gd, None


def edge_len(vec, mode=0, cnt=None):
    return _EdgeLen.apply(vec, mode, cnt)


def _torsion_seg(g, targ):
    """CSR over edges of the triplets whose torsion takes that edge's vector as its third argument."""
    T, E = g.T, g.E
    key = torch.empty(max(T, 1), dtype=torch.int32, device=targ.device)[:T]
    call('dig3d_torsion_key', ptr(targ), ptr(g.kj), ptr(getattr(g, 'val', None)), T, E, ptr(key), ptr(g.cnt_T), _stream())
    return key, csr_by_key(key, _hip.query('dig3d_torsion_key_segments', E))


class _TripGeom(Function):
    """(angle[, torsion]) of every triplet from the edge vectors.  VALUES come from the bit-exact float32 kernel on the
    positions (csrc/geometry.hip:k_triplet_geom, which also finds the torsion arg-min neighbour); the derivatives w.r.t.
    ``vec`` from csrc/diffgeom.hip."""

    @staticmethod
    def forward(ctx, vec, posd, g, use_torsion):
        from . import ops
        vec = _c(vec)
        angle, torsion, targ = ops.triplet_geom(posd, g, use_torsion)
        ctx.g, ctx.tor, ctx.targ = g, bool(use_torsion), targ
        ctx.save_for_backward(vec)
        return (angle, torsion) if use_torsion else angle

    @staticmethod
    def backward(ctx, *grads):
        (vec,) = ctx.saved_tensors
        g = ctx.g
        key = seg3 = None
        if ctx.tor:
            key, seg3 = _torsion_seg(g, ctx.targ)
        gt = grads[1] if ctx.tor else None
        return _TripGeomBwd.apply(vec, grads[0], gt, g, key, seg3), None, None, None


def _trip_pass(vec, gg, ga, gt, g, key, want_j):
    T, E = g.T, g.E
    dev = vec.device
    f = dict(dtype=torch.float32, device=dev)
    gv1 = torch.empty(max(T, 1), 3, **f)[:T]
    gv2 = torch.empty(max(T, 1), 3, **f)[:T]
    gv3 = torch.empty(max(T, 1), 3, **f)[:T] if key is not None else None
    o_ga = torch.empty(max(T, 1), **f)[:T] if want_j else None
    o_gt = torch.empty(max(T, 1), **f)[:T] if (want_j and key is not None) else None
    call('dig3d_tripgeom_grad', ptr(vec), ptr(gg), ptr(g.ji), ptr(g.kj), ptr(key), T, E, ptr(ga), ptr(gt), ptr(gv1),
         ptr(gv2), ptr(gv3), ptr(o_ga), ptr(o_gt), ptr(g.cnt_T), _stream())
    return gv1, gv2, gv3, o_ga, o_gt


class _TripGeomBwd(Function):
    @staticmethod
    def forward(ctx, vec, ga, gt, g, key, seg3):
        ga = _c(ga)
        gt = _c(gt) if gt is not None else None
        ctx.g, ctx.key, ctx.seg3 = g, key, seg3
        ctx.save_for_backward(vec, ga, gt)
        gv1, gv2, gv3, _, _ = _trip_pass(vec, None, ga, gt, g, key, False)
        return _combine(vec, None, None, g.E, g.tptr, gv1, g.seg_kj, gv2, seg3, gv3, cnt=g.cnt_E)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gg):
        vec, ga, gt = ctx.saved_tensors
        g, key, seg3 = ctx.g, ctx.key, ctx.seg3
        gg = _c(gg)
        hv1, hv2, hv3, o_ga, o_gt = _trip_pass(vec, gg, ga, gt, g, key, True)
        out, _ = _combine(vec, gg, None, g.E, g.tptr, hv1, g.seg_kj, hv2, seg3, hv3, cnt=g.cnt_E)
        return out, o_ga, o_gt, None, None, None


def triplet_angles(vec, posd, g, use_torsion):
    """-> angle or (angle, torsion), differentiable (twice) w.r.t. ``vec``."""
    return _TripGeom.apply(vec, posd, g, use_torsion)


def edge_vectors(pos, g):
    """vec[e] = pos[i] - pos[j] for edge e = (j -> i): two HIP row gathers (linear, closed under differentiation)."""
    from . import ops
    return ops.gather_rows(pos, g.seg_dst) - ops.gather_rows(pos, g.seg_src)


# ---------------------------------------------------------------------------------------------------------------
# basis
# ---------------------------------------------------------------------------------------------------------------
class _Bessel(Function):
    """bes[e, l*nr+n] = norm * j_l(z d/c) (* envelope) — csrc/basis.hip:k_bessel; derivatives csrc/diffgeom.hip."""

    @staticmethod
    def forward(ctx, dist, cutoff, ns, nr, zeros, norms, env_p, cnt):
        from . import ops
        dist = _c(dist)
        ctx.meta = (float(cutoff), int(ns), int(nr), zeros, norms, int(env_p), cnt)
        ctx.save_for_backward(dist)
        return ops.bessel_basis(dist, cutoff, ns, nr, zeros, norms, env_p)

    @staticmethod
    def backward(ctx, g):
        (dist,) = ctx.saved_tensors
        return (_BesselBwd.apply(dist, g, ctx.meta),) + (None,) * 7


def _bessel_grad(dist, g, gg, meta):
    cutoff, ns, nr, zeros, norms, env_p, cnt = meta
    E = dist.numel()
    o_d = torch.empty(E, dtype=torch.float32, device=dist.device)
    o_g = torch.empty(E, ns * nr, dtype=torch.float32, device=dist.device) if gg is not None else None
    call('dig3d_bessel_grad', ptr(dist), E, cutoff, ns, nr, ptr(zeros), ptr(norms), env_p, ptr(g), ptr(gg), ptr(o_d),
         ptr(o_g), ptr(cnt), _stream())
    return o_d, o_g


class _BesselBwd(Function):
    @staticmethod
    def forward(ctx, dist, g, meta):
        g = _c(g)
        ctx.meta = meta
        ctx.save_for_backward(dist, g)
        return _bessel_grad(dist, g, None, meta)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gg):
        dist, g = ctx.saved_tensors
        o_d, o_g = _bessel_grad(dist, g, _c(gg), ctx.meta)
        return o_d, o_g, None


def bessel_basis(dist, cutoff, ns, nr, zeros, norms, env_p=0, cnt=None):
    return _Bessel.apply(dist, cutoff, ns, nr, zeros, norms, env_p, cnt)


class _Harmonics(Function):
    """Y[m, h]: real spherical harmonics (h = l for phi None, else the ns^2 order of spherenet/features.py:249-250)."""

    @staticmethod
    def forward(ctx, theta, phi, ns, pref, cnt):
        theta = _c(theta)
        phi = _c(phi) if phi is not None else None
        M = theta.numel()
        H = ns if phi is None else ns * ns
        out = torch.empty(M, H, dtype=torch.float32, device=theta.device)
        call('dig3d_harmonics_fwd', ptr(theta), ptr(phi), M, int(ns), ptr(pref), ptr(out), ptr(cnt), _stream())
        ctx.meta = (int(ns), pref, cnt)
        ctx.save_for_backward(theta, phi)
        return out

    @staticmethod
    def backward(ctx, g):
        theta, phi = ctx.saved_tensors
        r = _HarmonicsBwd.apply(theta, phi, g, ctx.meta)
        if phi is None:
            return r, None, None, None, None
        return r[0], r[1], None, None, None


def _harm_grad(theta, phi, g, gg_th, gg_ph, order, meta):
    ns, pref, cnt = meta
    M = theta.numel()
    H = ns if phi is None else ns * ns
    f = dict(dtype=torch.float32, device=theta.device)
    o_th = torch.empty(M, **f)
    o_ph = torch.empty(M, **f) if phi is not None else None
    o_g = torch.empty(M, H, **f) if order == 2 else None
    call('dig3d_harmonics_grad', ptr(theta), ptr(phi), M, ns, ptr(pref), ptr(g), ptr(gg_th), ptr(gg_ph), order, ptr(o_th),
         ptr(o_ph), ptr(o_g), ptr(cnt), _stream())
    return o_th, o_ph, o_g


class _HarmonicsBwd(Function):
    @staticmethod
    def forward(ctx, theta, phi, g, meta):
        g = _c(g)
        ctx.meta = meta
        ctx.save_for_backward(theta, phi, g)
        o_th, o_ph, _ = _harm_grad(theta, phi, g, None, None, 1, meta)
        return o_th if phi is None else (o_th, o_ph)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gg):
        theta, phi, g = ctx.saved_tensors
        gg_th = _c(gg[0])
        gg_ph = _c(gg[1]) if phi is not None else None
        o_th, o_ph, o_g = _harm_grad(theta, phi, g, gg_th, gg_ph, 2, ctx.meta)
        return o_th, o_ph, o_g, None


def harmonics(theta, phi, ns, pref, cnt=None):
    return _Harmonics.apply(theta, phi, ns, pref, cnt)


class _DistEmb(Function):
    """rbf[e, n] = Envelope(d/c) sin(freq[n] d/c), freq learnable (spherenet/features.py:151-182) — one kernel forward,
    one backward (d and freq gradients), one for the double backward."""

    @staticmethod
    def forward(ctx, dist, freq, cutoff, p, cnt):
        dist, freq = _c(dist), _c(freq)
        E, nr = dist.numel(), freq.numel()
        out = torch.empty(E, nr, dtype=torch.float32, device=dist.device)
        call('dig3d_distemb_fwd', ptr(dist), ptr(freq), E, nr, float(cutoff), int(p), ptr(out), ptr(cnt), _stream())
        ctx.meta = (float(cutoff), int(p), cnt)
        # energy-only route with a leaf freq: its gradient's column sum joins the step's one reduction launch
        from . import ops
        ctx.leaf = bool(freq.is_leaf and not ops._twice_differentiable and not dist.requires_grad)
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.save_for_backward(dist, freq)
        return out

    @staticmethod
    def backward(ctx, g):
        dist, freq = ctx.saved_tensors
        if torch.is_grad_enabled() and ctx.pos_only:
            # the force gradient (run.py:126) asks for positions only: no freq gradient, no reduction launch for it (the
            # documented restriction of the energy_and_force layers: _warn_skipped_wgrad outside force_gradient_scope)
            _warn_skipped_wgrad(ctx.needs_input_grad[1])
            g_d, _ = _DistEmbBwd.apply(dist, freq, g, ctx.meta, False, True)
            return g_d, None, None, None, None
        g_d, g_f = _DistEmbBwd.apply(dist, freq, g, ctx.meta, ctx.leaf, False, ctx.pos_only)
        return g_d, g_f, None, None, None


def _distemb_grad(dist, freq, g, gg_d, gg_f, order, meta, leaf=False, skip_f=False, keyed=False):
    """``skip_f``: the freq gradient is not wanted (its partial rows are written and dropped).  In the final pass of an
    energy_and_force step a leaf freq receives two contributions (first- and second-order node): inside a deferred_reductions
    block both join the step's one reduction under the parameter's key instead of a column-sum launch each."""
    cutoff, p, cnt = meta
    E, nr = dist.numel(), freq.numel()
    f = dict(dtype=torch.float32, device=dist.device)
    o_d = torch.empty(max(E, 1), **f)[:E]
    o_g = torch.empty(max(E, 1), nr, **f)[:E] if order == 2 else None
    from . import ops
    nb = _hip.query('dig3d_distemb_blocks', E)
    part = torch.empty(nb * nr, **f)
    o_f = torch.empty(nr, **f)
    d = ops._deferred
    if skip_f and E > 0:
        now, o_ret = 0, None
    elif (keyed and E > 0 and d is not None and freq.is_leaf and not torch.is_grad_enabled()):
        gwb = d.add_keyed(freq.data_ptr(), part, nb, nr, nr, dist.device)
        now, o_ret = 0, gwb
        if gwb is not None:
            o_f = gwb
    else:
        now = 1 if (E <= 0 or order != 1) else ops._reduce_later(part, nb, nr, o_f, leaf)
        o_ret = o_f
    call('dig3d_distemb_grad', ptr(dist), ptr(freq), E, nr, cutoff, p, ptr(g), ptr(gg_d), ptr(gg_f), order, ptr(o_d),
         ptr(o_g), ptr(part), ptr(o_f), ptr(cnt), now, _stream())
    return o_d, o_ret, o_g


class _DistEmbBwd(Function):
    @staticmethod
    def forward(ctx, dist, freq, g, meta, leaf=False, skip_f=False, keyed=False):
        g = _c(g)
        ctx.meta = meta
        ctx.save_for_backward(dist, freq, g)
        ctx.set_materialize_grads(False)
        o_d, o_f, _ = _distemb_grad(dist, freq, g, None, None, 1, meta, leaf, skip_f, keyed)
        return o_d, o_f                      # o_f None: not wanted, or a later contribution under the parameter's key

    @staticmethod
    @once_differentiable
    def backward(ctx, gg_d, gg_f):
        dist, freq, g = ctx.saved_tensors
        gg_d = _c(gg_d) if gg_d is not None else None
        gg_f = _c(gg_f) if gg_f is not None else None
        o_d, o_f, o_g = _distemb_grad(dist, freq, g, gg_d, gg_f, 2, ctx.meta, keyed=True)
        return o_d, o_f, o_g, None, None, None, None


class _EdgeFront(Function):
    """(dist, rbf, bes) of the energy route in ONE launch (csrc/diffgeom.hip:k_edge_front): the edge lengths
    (geometric_computing.py:25), dist_emb (features.py:151-182; gradient w.r.t. the learnable freq as ``_DistEmb``) and the
    Bessel table of the angle / torsion embeddings.  Positions carry no gradient on this route."""

    @staticmethod
    def forward(ctx, pos, freq, g, mode, cutoff_d, p, cutoff_b, ns, nr, zeros, norms, env_p):
        from . import ops
        pos, freq = _c(pos), _c(freq)
        E, nrd = g.E, freq.numel()
        f = dict(dtype=torch.float32, device=pos.device)
        dist, rbf, bes = torch.empty(E, **f), torch.empty(E, nrd, **f), torch.empty(E, ns * nr, **f)
        call('dig3d_edge_front', ptr(pos), ptr(g.src), ptr(g.dst), E, int(mode), ptr(g.cnt_E), 1.0, ptr(dist), ptr(freq), nrd,
             float(cutoff_d), int(p), ptr(rbf), float(cutoff_b), int(ns), int(nr), ptr(zeros), ptr(norms), int(env_p),
             ptr(bes), _stream())
        ctx.meta = (float(cutoff_d), int(p), g.cnt_E)
        ctx.leaf = bool(freq.is_leaf and not ops._twice_differentiable)
        ctx.save_for_backward(dist, freq)
        ctx.mark_non_differentiable(dist, bes)
        # (without this autograd hands backward zero tensors for the two non-differentiable outputs: two framework fill
        # launches inside the replayed step, [E] and [E, ns nr])
        ctx.set_materialize_grads(False)
        return dist, rbf, bes

    @staticmethod
    def backward(ctx, _gd, g, _gb):
        dist, freq = ctx.saved_tensors
        if g is None:
            return (None,) * 12
        _, g_f = _DistEmbBwd.apply(dist, freq, g, ctx.meta, ctx.leaf)
        return (None, g_f) + (None,) * 10


def edge_front(pos, freq, g, mode, cutoff_d, p, cutoff_b, ns, nr, zeros, norms, env_p):
    return _EdgeFront.apply(pos, freq, g, mode, cutoff_d, p, cutoff_b, ns, nr, zeros, norms, env_p)


def dist_emb(dist, freq, cutoff, p, cnt=None):
    return _DistEmb.apply(dist, freq, cutoff, p, cnt)


# ---------------------------------------------------------------------------------------------------------------
# dense layer  y = act(x W^T + b) (+ res),  twice differentiable
# ---------------------------------------------------------------------------------------------------------------
def _keyed_partials(weight, nb, stride, n, dev):
    """scratch + gradient buffer of one weight-gradient contribution.  -> (part, gwb, now, mine): ``gwb`` is where the
    reduction lands (this contribution's own buffer, or the buffer of the contribution that registered the same weight
    first inside a ``deferred_reductions`` block), ``now`` the reduce_now flag of the C call, ``mine`` whether this
    contribution returns the buffer to autograd (False: return None, the partials are added at flush)."""
    from . import ops
    part = torch.empty(nb * stride, dtype=torch.float32, device=dev)
    d = ops._deferred
    if d is None or not weight.is_leaf:
        # (a weight that is itself computed — e.g. the product W2 W1 of two bias-free Linears — is differentiated
        # further during this very backward pass: its gradient must be complete now, not at flush)
        return part, torch.empty(stride, dtype=torch.float32, device=dev), 1, True
    key = weight.data_ptr()
    gwb = d.add_keyed(key, part, nb, stride, n, dev)
    if gwb is not None:
        return part, gwb, 0, True
    return part, d.owner(key), 0, False


class _LinAct2(Function):
    """Forward: ONE MFMA kernel.  Returns (y, z): z, the pre-activation, is an OUTPUT so that the double backward can
    send its act'' term straight back to it.  Backward: if autograd is recording (create_graph=True: the force
    gradient) the input gradient is the differentiable ``_DgradAct``; otherwise the fused first-order dgrad+wgrad
    launch, with the gradient that reached z added inside the kernel's staging (no separate merge pass).

    Layers created while a model runs its ``energy_and_force`` forward (ops.composite_mode) skip their WEIGHT
    gradients in a create_graph backward: that pass is ``grad(out, pos, create_graph=True)`` (run.py:126) and asks
    for position gradients only — autograd cannot tell a custom Function which of its outputs the caller wants."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act):
        from . import ops
        x, weight = _c(x), _c(weight)
        M, K = x.shape
        N = weight.size(0)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        z = torch.empty_like(y) if act != ACT_NONE else None
        call('dig3d_linear_fwd', ptr(x), ptr(weight), ptr(bias), ptr(res.contiguous() if res is not None else None), M, K,
             N, act, ptr(y), ptr(z), _stream())
        ctx.act, ctx.has_bias, ctx.has_res = act, bias is not None, res is not None
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        if z is None:
            z = y.new_empty(0)
            ctx.mark_non_differentiable(z)
        ctx.save_for_backward(x, weight, z)
        return y, z

    @staticmethod
    def backward(ctx, gy, gz):
        x, weight, z = ctx.saved_tensors
        act = ctx.act
        M, K = x.shape
        N = weight.size(0)
        st = _stream()
        dev = x.device
        if gy is None and gz is None:
            return None, None, None, None, None
        gz_add = _c(gz) if (gz is not None and act != ACT_NONE) else None
        zarg = z if act != ACT_NONE else None
        want_x = ctx.needs_input_grad[0]
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gres = gy if ctx.has_res else None
        gx = gw = gb = None
        if torch.is_grad_enabled():          # create_graph=True: gradients as differentiable Functions
            gsrc, act_eff = (_c(gy) if gy is not None else None), act
            if gz_add is not None:           # rare here (third order): merge explicitly
                G = torch.empty(M, N, dtype=torch.float32, device=dev)
                call('dig3d_preact_merge', ptr(gsrc), ptr(z), ptr(gz_add), M * N, act, ptr(G), st)
                gsrc, zarg, act_eff = G, None, ACT_NONE
            if want_x:
                gx = _DgradAct.apply(gsrc, zarg, weight, act_eff)
            _warn_skipped_wgrad(want_w and ctx.pos_only and weight.is_leaf)
            if want_w and not ctx.pos_only:
                gw, gb = _WgradAct.apply(gsrc, zarg, x, act_eff)
                gb = gb if ctx.has_bias else None
            return gx, gw, gb, gres, None
        gsrc = _c(gy) if gy is not None else torch.zeros(M, N, dtype=torch.float32, device=dev)
        stride = N * K + N
        if want_x and want_w:
            gx = torch.empty_like(x)
            nb = _hip.query('dig3d_linear_bwd_workers', M, K, N)
            part, gwb, now, mine = _keyed_partials(weight, nb, stride, stride, dev)
            call('dig3d_linear_bwd_zadd', ptr(gsrc), ptr(zarg), ptr(weight), ptr(x), M, K, N, act, ptr(gx), None,
                 ptr(part), ptr(gwb), now, ptr(gz_add), st)
        elif want_x or want_w:
            if gz_add is not None:
                G = torch.empty(M, N, dtype=torch.float32, device=dev)
                call('dig3d_preact_merge', ptr(gsrc), ptr(z), ptr(gz_add), M * N, act, ptr(G), st)
                gsrc, zarg, act = G, None, ACT_NONE
            if want_x:
                gx = torch.empty_like(x)
                call('dig3d_linear_bwd_input', ptr(gsrc), ptr(zarg), ptr(weight), M, K, N, act, ptr(gx), None, st)
            else:
                nb = _hip.query('dig3d_linear_wgrad_blocks', M)
                part, gwb, now, mine = _keyed_partials(weight, nb, stride, stride, dev)
                call('dig3d_linear_bwd_weight', ptr(gsrc), ptr(zarg), ptr(x), M, K, N, act, ptr(part), ptr(gwb), now, st)
        if want_w:
            gw = gwb[:N * K].view(N, K) if mine else None
            gb = gwb[N * K:] if ctx.has_bias else None
        return gx, gw, gb, gres, None


class _DgradAct(Function):
    """gx = (gy * act'(z)) W — the input gradient of the dense layer as a differentiable function of (gy, z, W)."""

    @staticmethod
    def forward(ctx, gy, z, weight, act):
        gy = _c(gy)
        M, N = gy.shape
        K = weight.size(1)
        gx = torch.empty(M, K, dtype=torch.float32, device=gy.device)
        call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(weight), M, K, N, act, ptr(gx), None, _stream())
        ctx.act = act
        ctx.save_for_backward(gy, z, weight)
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx):
        gy, z, weight = ctx.saved_tensors
        act = ctx.act
        ggx = _c(ggx)
        M, N = gy.shape
        K = weight.size(1)
        st = _stream()
        dev = gy.device
        o_gy = o_z = gw = None
        if N > 64 and ctx.needs_input_grad[2] and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            # all three results in ONE launch: weight-gradient workers + row tiles with the second-order epilogue
            stride = N * K + N
            nb = _hip.query('dig3d_linear_dd_workers', M, K, N)
            part, gwb, now, mine = _keyed_partials(weight, nb, stride, N * K, dev)
            o_gy = torch.empty(M, N, dtype=torch.float32, device=dev)
            o_z = torch.empty_like(o_gy) if act != ACT_NONE else None
            call('dig3d_linear_dd', ptr(ggx), ptr(weight), ptr(z), ptr(gy), M, K, N, act, ptr(o_gy), ptr(o_z), ptr(part),
                 ptr(gwb), now, st)
            return o_gy, o_z, (gwb[:N * K].view(N, K) if mine else None), None
        if ctx.needs_input_grad[0] or (z is not None and ctx.needs_input_grad[1]):
            # t = ggx W^T on the MFMA kernel; its epilogue applies act' / act'' (no separate elementwise pass)
            o_gy = torch.empty(M, N, dtype=torch.float32, device=dev)
            if act == ACT_NONE:
                call('dig3d_linear_fwd', ptr(ggx), ptr(weight), None, None, M, K, N, ACT_NONE, ptr(o_gy), None, st)
            else:
                o_z = torch.empty_like(o_gy)
                call('dig3d_linear_fwd', ptr(ggx), ptr(weight), ptr(z), ptr(gy), M, K, N, 8 + act, ptr(o_gy), ptr(o_z), st)
        if ctx.needs_input_grad[2]:
            stride = N * K + N
            nb = _hip.query('dig3d_linear_wgrad_blocks', M)
            part, gwb, now, mine = _keyed_partials(weight, nb, stride, N * K, dev)
            call('dig3d_linear_bwd_weight', ptr(gy), ptr(z), ptr(ggx), M, K, N, act, ptr(part), ptr(gwb), now, st)
            gw = gwb[:N * K].view(N, K) if mine else None
        return o_gy, o_z, gw, None


class _WgradAct(Function):
    """(gW, gb) = ((gy * act'(z))^T x, column sums) — the weight gradient as a differentiable function of (gy, z, x)."""

    @staticmethod
    def forward(ctx, gy, z, x, act):
        gy, x = _c(gy), _c(x)
        M, N = gy.shape
        K = x.size(1)
        stride = N * K + N
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        part = torch.empty(nb * stride, dtype=torch.float32, device=gy.device)
        gwb = torch.empty(stride, dtype=torch.float32, device=gy.device)
        call('dig3d_linear_bwd_weight', ptr(gy), ptr(z), ptr(x), M, K, N, act, ptr(part), ptr(gwb), 1, _stream())
        ctx.act = act
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(gy, z, x)
        return gwb[:N * K].view(N, K), gwb[N * K:]

    @staticmethod
    @once_differentiable
    def backward(ctx, ggw, ggb):
        gy, z, x = ctx.saved_tensors
        act = ctx.act
        M, N = gy.shape
        K = x.size(1)
        st = _stream()
        dev = gy.device
        if ggw is None and ggb is None:
            return None, None, None, None
        ggw = _c(ggw) if ggw is not None else torch.zeros(N, K, dtype=torch.float32, device=dev)
        ggb = _c(ggb) if ggb is not None else None
        # d/d(gz): x ggW^T + ggb   (gz = gy act'(z))
        t = torch.empty(M, N, dtype=torch.float32, device=dev)
        call('dig3d_linear_fwd', ptr(x), ptr(ggw), ptr(ggb), None, M, K, N, ACT_NONE, ptr(t), None, st)
        o_z = None
        if act == ACT_NONE:
            o_gy = t
        else:
            o_gy = torch.empty_like(t)
            o_z = torch.empty_like(t)
            call('dig3d_act_bwd2', ptr(t), ptr(gy), ptr(z), M * N, act, ptr(o_gy), ptr(o_z), st)
        gx = None
        if ctx.needs_input_grad[2]:
            gx = torch.empty(M, K, dtype=torch.float32, device=dev)
            call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(ggw), M, K, N, act, ptr(gx), None, st)
        return o_gy, o_z, gx, None


def linear2(x, weight, bias=None, act=ACT_NONE, res=None):
    """act(F.linear(x, weight, bias)) (+ res), differentiable twice."""
    return _LinAct2.apply(x, weight, bias, res, act)[0]


# ---------------------------------------------------------------------------------------------------------------
# gather-multiply-aggregate, closed under differentiation
#     F(X, A; gat, seg)[s] = sum_{t in seg(s)} X[gat.key[t]] * A[t]         (csrc/segment.hip:k_seg_fused)
#     P(G, X; ig, ix)[t]   = G[ig.key[t]] * X[ix.key[t]]                    (k_gather_mul2)
#   dF/dX = F(g, A; seg, gat)   dF/dA = P(g, X; seg, gat)   dP/dG = F(X, h; ix, ig)   dP/dX = F(G, h; ig, ix)
# ---------------------------------------------------------------------------------------------------------------
class _GMS(Function):
    @staticmethod
    def forward(ctx, X, A, gat, seg):
        from . import ops
        X, A = _c(X), _c(A)
        ctx.gat, ctx.seg = gat, seg
        ctx.save_for_backward(X, A)
        return ops.segment_fused_raw(X, gat.key, A, None, seg, X.size(1))

    @staticmethod
    def backward(ctx, g):
        X, A = ctx.saved_tensors
        gX = _GMS.apply(g, A, ctx.seg, ctx.gat) if ctx.needs_input_grad[0] else None
        gA = _GM2.apply(g, X, ctx.seg, ctx.gat) if ctx.needs_input_grad[1] else None
        return gX, gA, None, None


class _GM2(Function):
    @staticmethod
    def forward(ctx, G, X, ig, ix):
        G, X = _c(G), _c(X)
        M, C = ig.key.numel(), X.size(1)
        out = torch.empty(M, C, dtype=torch.float32, device=X.device)
        call('dig3d_gather_mul2', ptr(G), ptr(ig.key), ptr(X), ptr(ix.key), None, None, M, C, ptr(out), None,
             ptr(ig.cnt if ig.cnt is not None else ix.cnt), _stream())
        ctx.ig, ctx.ix = ig, ix
        ctx.save_for_backward(G, X)
        return out

    @staticmethod
    def backward(ctx, h):
        G, X = ctx.saved_tensors
        gG = _GMS.apply(X, h, ctx.ix, ctx.ig) if ctx.needs_input_grad[0] else None
        gX = _GMS.apply(G, h, ctx.ig, ctx.ix) if ctx.needs_input_grad[1] else None
        return gG, gX, None, None


def gather_mul_segsum(X, A, gat, seg):
    """sum_{t in seg(s)} X[gat.key[t]] * A[t] — differentiable to any order on two kernels."""
    return _GMS.apply(X, A, gat, seg)


# ---------------------------------------------------------------------------------------------------------------
# fused triplet interaction WITHOUT a torsion factor (DimeNet++, dimenetpp.py:146-150), closed under differentiation on
# the energy route's kernels (csrc/triplet.hip / triplet_wave.hip) — round 5.
#     T(X, W, P)[e, c] = sum_{t: ji[t] = e} X[kj[t], c] * sum_b W[c, b] P[t, b]          dig3d_triplet_fwd
# is TRILINEAR, and its three partial gradients are trilinear forms of the same family:
#     A(G, W, P)[r, c] = sum_{t: kj[t] = r} G[ji[t], c] * sum_b W[c, b] P[t, b]          dig3d_triplet_fwd, transposed CSR
#     B(G, X, W)[t, b] = sum_c G[ji[t], c] X[kj[t], c] W[c, b]                           dig3d_triplet_bwd (gPs)
#     C(G, X, P)[c, b] = sum_t G[ji[t], c] X[kj[t], c] P[t, b]                           dig3d_triplet_bwd (gW2s), same launch
#   dT = (A(g,W,P), C(g,X,P), B(g,X,W))     dA/d(G,W,P) . h = (T(h,W,P), C(G,h,P), B(G,h,W))
#   d(B,C)/d(G,X,W,P) . (q, w) = (T(X,W,q) + T(X,w,P),  A(G,W,q) + A(G,w,P),  C(G,X,q),  B(G,X,w))
# so the energy_and_force step (forward, create_graph backward, its backward, final backward) stays on three launches of
# two kernels and never forms the [T, int_emb] factor tensor the table route needs: sbf -> lin_sbf1 -> P [T, 8] is the
# only T-sized product (was: lin_sbf2 lin_sbf1 as ONE [T, 42] -> [T, 64] layer per block = 16 x k_linear_fwd<2>,
# 27 x k_linear_bwd_both / k_linear_bwd_input_s at T rows, 24 x k_seg_fused<16>, 12 x k_gather_mul2 per config-3 step).
# W: [C, 8] (zero-padded columns), P: [T, 8]; C in {16, 32, 64, 128, 256}.
# ---------------------------------------------------------------------------------------------------------------
_in_force_gradient = 0      # > 0 inside ``force_gradient`` (the one create_graph backward that needs no weight gradient)
_warned_wgrad = False


class force_gradient_scope:
    """``with force_gradient_scope(): grad(out, pos, create_graph=True)`` — run.py:126.  Layers built inside an
    energy_and_force forward skip their WEIGHT gradients in a create_graph backward (autograd cannot tell a custom Function which
    of its inputs the caller asked for); this scope says that the skip is intended.  A create_graph backward OUTSIDE it that
    reaches a leaf weight (Hessian-vector products, meta-gradients) warns once instead of returning None silently."""

    def __enter__(self):
        global _in_force_gradient
        _in_force_gradient += 1

    def __exit__(self, *a):
        global _in_force_gradient
        _in_force_gradient -= 1


def _warn_skipped_wgrad(skipped):
    global _warned_wgrad
    if skipped and not _in_force_gradient and not _warned_wgrad:
        import warnings
        warnings.warn('dig_amd: a create_graph backward through an energy_and_force forward does not produce WEIGHT gradients '
                      '(only the position gradient of run.py:126 is supported at second order); wrap the intended force-'
                      'gradient call in dig_amd.diffops.force_gradient_scope() to silence this')
        _warned_wgrad = True


def _trip_route():
    from . import ops
    return int(ops.trip_lane_groups)


def _trip_T_raw(X, W, P, g, add=None):
    """``add`` [E, C]: added to the result INSIDE the launch (the other gradient reaching the same tensor in the final pass)."""
    E, C = X.shape
    if P.size(0) == 0 or E == 0:
        z = torch.zeros(E, C, dtype=torch.float32, device=X.device)
        return z if add is None else z + add
    out = torch.empty(E, C, dtype=torch.float32, device=X.device)
    call('dig3d_triplet_fwd_add', ptr(X), ptr(g.kj), ptr(P), None, ptr(W), None, ptr(g.tptr), None, E, C, ptr(out),
         ptr(_c(add)) if add is not None else None, _trip_route(), _stream())
    return out


def _trip_A_raw(G, W, P, g, add=None):
    E, C = G.shape
    if P.size(0) == 0 or E == 0:
        z = torch.zeros(E, C, dtype=torch.float32, device=G.device)
        return z if add is None else z + add
    seg = g.seg_kj
    out = torch.empty(E, C, dtype=torch.float32, device=G.device)
    call('dig3d_triplet_fwd_add', ptr(G), ptr(g.ji), ptr(P), None, ptr(W), None, ptr(seg.kptr), ptr(seg.perm), E, C, ptr(out),
         ptr(_c(add)) if add is not None else None, _trip_route(), _stream())
    return out


def _trip_BC_raw(G, X, W, P, g, want_c=True, wkey=None, gp_add=None):
    """-> (B(G, X, W) [T, 8], C(G, X, P) [C, 8] or None) in one launch.  ``want_c=False``: the block partials of C are
    written and dropped (no reduction launch).  ``wkey``: the weight this C is a gradient contribution of — inside a
    ``deferred_reductions`` block its partials join the step's ONE reduction (a weight enters the second-order graph three
    times: only the first contribution hands a buffer to autograd, see ``_keyed_partials``); None: reduced here."""
    E, C = X.shape
    T = P.size(0)
    dev = X.device
    if T == 0 or E == 0:
        z = torch.zeros(T, 8, dtype=torch.float32, device=dev)
        return (z if gp_add is None else z + gp_add,
                (torch.zeros(C, 8, dtype=torch.float32, device=dev) if want_c else None))
    gP = torch.empty(T, 8, dtype=torch.float32, device=dev)
    route = _trip_route()
    nb = _hip.query('dig3d_triplet_bwd_blocks', E, C, route)
    stride = 2 * C * 8
    if want_c and wkey is not None:
        part, gwb, now, mine = _keyed_partials(wkey, nb, stride, C * 8, dev)
    else:
        part = torch.empty(nb * stride, dtype=torch.float32, device=dev)
        gwb, now, mine = torch.empty(stride, dtype=torch.float32, device=dev), (1 if want_c else 0), want_c
    call('dig3d_triplet_bwd_add', ptr(G), ptr(X), ptr(g.kj), ptr(P), None, ptr(W), None, ptr(g.tptr), E, C, ptr(gP), None,
         ptr(part), ptr(gwb), None, now, route, ptr(_c(gp_add)) if gp_add is not None else None, None, _stream())
    if g.cnt_T is not None and getattr(g, 'zero_trip_tail', True):
        # padded triplets of a static-shape batch belong to no segment: their rows are never written, and the dense layer
        # that consumes gP walks every row -> zeros behind the live count (was a 4-MB fill of the whole buffer per call,
        # 16 per config-3 step).  (Not needed when the consumer is the fused basis projection: csrc/sbf2.hip never reads a row
        # behind the live count.)
        call('dig3d_zero_rows_from', ptr(gP), g.cnt_T.data_ptr(), T, 8, _stream())
    return gP, (gwb[:C * 8].view(C, 8) if mine else None)


class _TripT(Function):
    """(T(X, W, P), X', P'): the aliases are what the create_graph backward of an energy_and_force forward differentiates
    through (``_TripBwd2``, ONE Function for its two launches), so that in the final pass G, X and P each have a single consumer
    and what ``_TripBwd2.backward`` sends to X' and P' arrives HERE and is added inside the A and B launches (the pattern of
    ``_Mul2``; tools/diag_force_fanin.py: 4 additions per block and step on [E, 64], [E, 64] and twice [T, 8])."""

    @staticmethod
    def forward(ctx, X, W, P, g):
        from . import ops
        X, W, P = _c(X), _c(W), _c(P)
        ctx.g = g
        # a forward inside a model's energy_and_force pass: its create_graph backward is the POSITION gradient, which needs
        # no weight gradient (the documented restriction of the twice-differentiable dense layers, _LinAct2)
        ctx.pos_only = bool(ops._twice_differentiable)
        X2, P2 = X.view_as(X), P.view_as(P)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(X, W, P, X2, P2)
        return _trip_T_raw(X, W, P, g), X2, P2

    @staticmethod
    def backward(ctx, G, gX2, gP2):
        X, W, P, X2, P2 = ctx.saved_tensors
        g = ctx.g
        if G is None:
            return gX2, None, gP2, None
        live = torch.is_grad_enabled()
        want_c = ctx.needs_input_grad[1] and not (ctx.pos_only and live)
        _warn_skipped_wgrad(ctx.needs_input_grad[1] and not want_c)
        if live and ctx.pos_only and not want_c:
            # the force gradient: gX = A(G, W, P), gP = B(G, X, W) as one differentiable Function of (G, X', W, P')
            gX, gP = _TripBwd2.apply(G, X2, W, P2, g)
            return (gX if gX2 is None else gX + gX2), None, (gP if gP2 is None else gP + gP2), None
        if not live:
            # final pass: what this op's own create_graph backward sent to X' / P' is added inside the launches
            gX = _trip_A_raw(_c(G), W, P, g, add=gX2) if ctx.needs_input_grad[0] else None
            gP = gW = None
            if want_c or ctx.needs_input_grad[2]:
                gP, gW = _trip_BC_raw(_c(G), X, W, P, g, want_c, W if want_c else None, gp_add=gP2)
            return gX, gW, gP, None
        gX = _TripA.apply(G, W, P, g, ctx.pos_only) if ctx.needs_input_grad[0] else None
        gP = gW = None
        if want_c or ctx.needs_input_grad[2]:
            gP, gW = _TripBC.apply(G, X, W, P, g, want_c)
        if gX2 is not None:
            gX = gX2 if gX is None else gX + gX2
        if gP2 is not None:
            gP = gP2 if gP is None else gP + gP2
        return gX, gW, gP, None


class _TripBwd2(Function):
    """(gX, gP) = (A(G, W, P), B(G, X, W)): the create_graph backward of ``_TripT`` on the energy_and_force route (no weight
    gradient there).  Its backward is the final pass: with H = d/d gX, Q = d/d gP
        gG = T(H, W, P) + T(X, W, Q)     (the sum inside the second launch)
        gX = A(G, W, Q),  gP = B(G, H, W),  gW = C(G, H, P) + C(G, X, Q)  (keyed contributions of the deferred reduction)."""

    @staticmethod
    def forward(ctx, G, X, W, P, g):
        G, X, W, P = _c(G), _c(X), _c(W), _c(P)
        ctx.g = g
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(G, X, W, P)
        gX = _trip_A_raw(G, W, P, g)
        gP, _ = _trip_BC_raw(G, X, W, P, g, False)
        return gX, gP

    @staticmethod
    def backward(ctx, H, Q):
        G, X, W, P = ctx.saved_tensors
        g = ctx.g
        if H is None and Q is None:
            return None, None, None, None, None
        if torch.is_grad_enabled():              # a third-order pass: compose the open family (framework additions)
            gG = gXi = gW = gPi = None
            if H is not None:
                gG = _TripT.apply(H, W, P, g)[0]
                gPi, gW = _TripBC.apply(G, H, W, P, g, ctx.needs_input_grad[2])
            if Q is not None:
                t = _TripT.apply(X, W, Q, g)[0]
                gG = t if gG is None else gG + t
                gXi = _TripA.apply(G, W, Q, g)
                if ctx.needs_input_grad[2]:
                    _, w2 = _TripBC.apply(G, X, W, Q, g, True)
                    gW = w2 if gW is None else gW + w2
            return gG, gXi, gW, gPi, None
        want_w = ctx.needs_input_grad[2]
        gG = gXi = gPi = None
        gWs = []
        if H is not None:
            H = _c(H)
            if ctx.needs_input_grad[0]:
                gG = _trip_T_raw(H, W, P, g)
            if want_w or ctx.needs_input_grad[3]:
                gPi, w1 = _trip_BC_raw(G, H, W, P, g, want_w, W if want_w else None)      # B(G, H, W), C(G, H, P)
                gWs.append(w1)
        if Q is not None:
            Q = _c(Q)
            if ctx.needs_input_grad[0]:
                gG = _trip_T_raw(X, W, Q, g, add=gG)
            if ctx.needs_input_grad[1]:
                gXi = _trip_A_raw(G, W, Q, g)
            if want_w:
                _, w2 = _trip_BC_raw(G, X, W, Q, g, True, W)                               # C(G, X, Q)
                gWs.append(w2)
        gWs = [w for w in gWs if w is not None]
        gW = None
        if gWs:
            gW = gWs[0]
            for w in gWs[1:]:
                gW = gW + w
        return gG, gXi, gW, gPi, None


class _TripA(Function):
    @staticmethod
    def forward(ctx, G, W, P, g, pos_only=False):
        G, W, P = _c(G), _c(W), _c(P)
        ctx.g, ctx.pos_only = g, pos_only
        ctx.save_for_backward(G, W, P)
        return _trip_A_raw(G, W, P, g)

    @staticmethod
    def backward(ctx, H):
        G, W, P = ctx.saved_tensors
        g = ctx.g
        want_c = ctx.needs_input_grad[1] and not (ctx.pos_only and torch.is_grad_enabled())
        gG = _TripT.apply(H, W, P, g)[0] if ctx.needs_input_grad[0] else None
        gP = gW = None
        if want_c or ctx.needs_input_grad[2]:
            gP, gW = _TripBC.apply(G, H, W, P, g, want_c)
        return gG, gW, gP, None, None


class _TripBC(Function):
    @staticmethod
    def forward(ctx, G, X, W, P, g, want_c=True):
        G, X, W, P = _c(G), _c(X), _c(W), _c(P)
        ctx.g = g
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(G, X, W, P)
        # (W is the weight C(G, X, P) is a gradient of — the key of its deferred reduction)
        gP, gW = _trip_BC_raw(G, X, W, P, g, want_c, W if want_c else None)
        return gP, gW

    @staticmethod
    def backward(ctx, Q, Wh):
        G, X, W, P = ctx.saved_tensors
        g = ctx.g
        gG = gX = gW = gP = None
        if Q is not None:                       # through B(G, X, W)
            if ctx.needs_input_grad[0]:
                gG = _TripT.apply(X, W, Q, g)[0]
            if ctx.needs_input_grad[1]:
                gX = _TripA.apply(G, W, Q, g)
            if ctx.needs_input_grad[2]:
                _, gW = _TripBC.apply(G, X, W, Q, g, True)       # C(G, X, Q): a gradient of W
        if Wh is not None:                      # through C(G, X, P)
            if ctx.needs_input_grad[0]:
                t = _TripT.apply(X, Wh, P, g)[0]
                gG = t if gG is None else gG + t
            if ctx.needs_input_grad[1]:
                t = _TripA.apply(G, Wh, P, g)
                gX = t if gX is None else gX + t
            if ctx.needs_input_grad[3]:
                gP, _ = _TripBC.apply(G, X, Wh, P, g, False)     # B(G, X, Wh)
        return gG, gX, gW, gP, None, None


def trip2_supported(X, P, W):
    """shapes the closed triplet family covers: X [E, C] with C a channel count of the fused kernels, P [T, bs], W [C, bs],
    bs <= 8"""
    return (X.is_cuda and X.dim() == 2 and X.dtype == torch.float32 and X.size(1) in (16, 32, 64, 128, 256) and P.dim() == 2
            and W.dim() == 2 and W.size(0) == X.size(1) and P.size(1) == W.size(1) <= 8)


def trip2_shapes_ok(X, bs, W):
    """the same check before P exists: ``bs`` = width of the projected basis (``lin_sbf1.out_features``)"""
    return (X.is_cuda and X.dim() == 2 and X.dtype == torch.float32 and X.size(1) in (16, 32, 64, 128, 256)
            and W.dim() == 2 and tuple(W.shape) == (X.size(1), bs) and bs <= 8)


def trip2(X, P, W, g):
    """sum_{t: ji[t] = e} X[kj[t]] * (W P[t])  — ``x_kj[idx_kj] * lin_sbf2(P)`` + ``scatter(..., idx_ji)``
    (dimenetpp.py:146-150) with P = lin_sbf1(sbf) [T, bs]; differentiable to any order."""
    from . import ops
    bs = P.size(1)
    if bs < 8:                                  # the kernels read 8-wide rows: zero columns (closed under differentiation)
        P = ops.pad2d(P, P.size(0), 8)
        W = ops.pad2d(W, W.size(0), 8)
    return _TripT.apply(X, W, P, g)[0]


# ---------------------------------------------------------------------------------------------------------------
# The angular basis contracted with lin_sbf1 of ALL blocks (dimenetpp/features.py:183-220 + dimenetpp.py:146), closed under
# differentiation on two kernels (csrc/sbf2.hip): the [T, ns nr] table is never formed, seven launches per step.
#     P_b[t, c] = sum_ln W[8 b + c, ln] bes[kj[t], ln] Y_l(angle[t])
# ---------------------------------------------------------------------------------------------------------------
def _sbf_t(A, B, s, angle, g, W, meta, gPs, want_p, want_a, want_w, want_h):
    ns, nr, pref = meta
    T, J = angle.numel(), W.size(0)
    dev = angle.device
    f = dict(dtype=torch.float32, device=dev)
    outP = [torch.empty(max(T, 1), 8, **f)[:T] for _ in range(J // 8)] if want_p else None
    outA = torch.empty(max(T, 1), **f)[:T] if want_a else None
    H = torch.empty(max(T, 1), 8, **f)[:T] if want_h else None
    part = wrow = None
    if want_w:
        nb = _hip.query('dig3d_sbf2_blocks', T)
        wrow = _keyed_partials(W, nb, J * ns * nr, J * ns * nr, dev)
        part = wrow[0]
    if T > 0:
        pg, k1 = _ptr_arr(gPs) if gPs is not None else (None, None)
        po, k2 = _ptr_arr(outP) if outP is not None else (None, None)
        call('dig3d_sbf2_t', ptr(A), ptr(B), ptr(s), ptr(angle), ptr(g.kj), ptr(W), J, ns, nr, ptr(pref), pg, po, ptr(outA),
             ptr(part), ptr(H), T, ptr(g.cnt_T), _stream())
        if want_w and wrow[2]:                  # no deferred reduction open: reduce here
            import ctypes
            cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
            n = J * ns * nr
            call('dig3d_reduce_many', cast((ctypes.c_void_p * 1)(ptr(part))), cast((ctypes.c_int * 1)(nb)),
                 cast((ctypes.c_int64 * 1)(n)), cast((ctypes.c_int * 1)(n)), cast((ctypes.c_void_p * 1)(ptr(wrow[1]))), 1, _stream())
    elif want_w:
        wrow[1].zero_()
    gW = (wrow[1].view(J, ns * nr) if wrow[3] else None) if want_w else None
    return outP, outA, gW, H


def _sbf_e(H, gPs, W, meta, g):
    ns, nr, _ = meta
    E, J = g.E, W.size(0)
    out = torch.empty(max(E, 1), ns * nr, dtype=torch.float32, device=W.device)[:E]
    if E > 0 and H.size(0) > 0:
        seg = g.seg_kj
        pg, k1 = _ptr_arr(gPs)
        call('dig3d_sbf2_e', ptr(H), pg, ptr(W), J, ns, nr, ptr(seg.kptr), ptr(seg.perm), E, ptr(out), _stream())
    else:
        out.zero_()
    return out


class _SbfProj(Function):
    """(P_0 .. P_{L-1}) [T, 8] each = the stacked projection of the angular basis; W [8 L, ns nr] (a leaf)."""

    @staticmethod
    def forward(ctx, bes, angle, W, g, meta):
        from . import ops
        bes, angle, W = _c(bes), _c(angle), _c(W)
        ctx.g, ctx.meta = g, meta
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(bes, angle, W)
        outP, _, _, _ = _sbf_t(bes, None, None, angle, g, W, meta, None, True, False, False, False)
        return tuple(outP)

    @staticmethod
    def backward(ctx, *gPs):
        bes, angle, W = ctx.saved_tensors
        if all(gp is None for gp in gPs):
            return None, None, None, None, None
        T = angle.numel()
        gPs = [_c(gp) if gp is not None else torch.zeros(T, 8, dtype=torch.float32, device=angle.device) for gp in gPs]
        want_w = ctx.needs_input_grad[2] and not (ctx.pos_only and torch.is_grad_enabled())
        g_bes, g_a, gW = _SbfBwd.apply(bes, angle, W, ctx.g, ctx.meta, want_w, *gPs)
        return g_bes, g_a, gW, None, None


class _SbfBwd(Function):
    """(g_bes, g_angle, gW) of _SbfProj as a differentiable function of (bes, angle, W, gP_0 ..)."""

    @staticmethod
    def forward(ctx, bes, angle, W, g, meta, want_w, *gPs):
        gPs = [_c(gp) for gp in gPs]
        ctx.g, ctx.meta = g, meta
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(bes, angle, W, *gPs)
        _, g_a, gW, H = _sbf_t(bes, None, None, angle, g, W, meta, gPs, False, True, want_w, True)
        g_bes = _sbf_e(H, gPs, W, meta, g)
        if gW is None:
            return g_bes, g_a, None
        return g_bes, g_a, gW

    @staticmethod
    @once_differentiable
    def backward(ctx, c_bes, c_a, c_w):
        sv = ctx.saved_tensors
        bes, angle, W, gPs = sv[0], sv[1], sv[2], list(sv[3:])
        g, meta = ctx.g, ctx.meta
        L = len(gPs)
        if c_w is not None:
            raise NotImplementedError('dig_amd sbf projection: third-order differentiation is not supported')
        if c_bes is None and c_a is None:
            return (None,) * (6 + L)
        E, K = bes.shape
        T = angle.numel()
        dev = bes.device
        # d/d(.) of  <c_bes, g_bes> + <c_a, g_angle>:  one thread-per-triplet launch with A = c_bes, B = bes, s = c_a
        A = _c(c_bes) if c_bes is not None else torch.zeros(E, K, dtype=torch.float32, device=dev)
        if c_a is not None:
            dP, d_a, dW, H = _sbf_t(A, bes, _c(c_a), angle, g, W, meta, gPs, True, True, ctx.needs_input_grad[2], True)
            d_bes = _sbf_e(H, gPs, W, meta, g)
        else:
            dP, d_a, dW, H = _sbf_t(A, None, None, angle, g, W, meta, gPs, True, True, ctx.needs_input_grad[2], False)
            d_bes = None
        return (d_bes, d_a, dW, None, None, None) + tuple(dP)


def sbf_project_supported(ns, nr, L, bs):
    return L in (1, 2, 4, 8) and all(b == 8 for b in bs) and bool(_hip.query('dig3d_sbf2_supported', int(ns), int(nr)))


def sbf_project(bes, angle, Wstack, g, ns, nr, pref):
    """[lin_sbf1_b(bes[idx_kj] (x) Y(angle)) for every block b] — L tensors [T, 8], twice differentiable w.r.t. (bes, angle),
    once more w.r.t. the stacked weight ``Wstack`` [8 L, ns nr]."""
    return list(_SbfProj.apply(bes, angle, Wstack, g, (int(ns), int(nr), pref)))


class _SplitCols8(Function):
    """x [T, 8 L] -> L contiguous [T, 8] tensors (csrc/readout.hip:k_cols_split8); the adjoint is _MergeCols8."""

    @staticmethod
    def forward(ctx, x, L):
        import ctypes
        x = _c(x)
        T = x.size(0)
        outs = [torch.empty(T, 8, dtype=torch.float32, device=x.device) for _ in range(L)]
        arr = (ctypes.c_void_p * L)(*[ptr(o) for o in outs])
        call('dig3d_cols_split8', ptr(x), T, L, ctypes.cast(arr, ctypes.c_void_p), _stream())
        ctx.L = L
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        T = next(g for g in gs if g is not None).size(0)
        return _MergeCols8.apply(ctx.L, T, *gs), None


class _MergeCols8(Function):
    @staticmethod
    def forward(ctx, L, T, *xs):
        import ctypes
        xs = [(_c(x) if x is not None else None) for x in xs]
        dev = next(x for x in xs if x is not None).device
        out = torch.empty(T, 8 * L, dtype=torch.float32, device=dev)
        arr = (ctypes.c_void_p * L)(*[ptr(x) for x in xs])
        call('dig3d_cols_merge8', ctypes.cast(arr, ctypes.c_void_p), T, L, ptr(out), _stream())
        ctx.L, ctx.has = L, [x is not None for x in xs]
        return out

    @staticmethod
    def backward(ctx, g):
        parts = _SplitCols8.apply(g, ctx.L)
        return (None, None) + tuple(p if h else None for p, h in zip(parts, ctx.has))


def split_cols8(x, L):
    """[T, 8 L] -> tuple of L contiguous [T, 8] column groups; linear, closed under differentiation"""
    return _SplitCols8.apply(x, L)


# ---------------------------------------------------------------------------------------------------------------
# grouped dense layers, twice differentiable: the L + 1 output blocks of the energy_and_force route (every stage of all
# blocks in one launch, as dig_amd/ops.py:grouped_readout does for the energy-only route)
# ---------------------------------------------------------------------------------------------------------------
def _ptrs(ts):
    import ctypes
    arr = (ctypes.c_void_p * len(ts))(*[ptr(t) for t in ts])
    return ctypes.cast(arr, ctypes.c_void_p), arr


class _GroupedLinAct2(Function):
    """(ys, zs) = G same-shape layers y_g = act(x_g W_g^T + b_g); position-only create_graph backward (see _LinAct2)."""

    @staticmethod
    def forward(ctx, act, G, *tensors):
        xs = [_c(t) for t in tensors[:G]]
        Ws = [_c(t) for t in tensors[G:2 * G]]
        bs = list(tensors[2 * G:3 * G])
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        ys = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)]
        zs = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)] if act != ACT_NONE else None
        px, k1 = _ptrs(xs)
        pw, k2 = _ptrs(Ws)
        pb, k3 = _ptrs(bs)
        py, k4 = _ptrs(ys)
        pz, k5 = _ptrs(zs) if zs is not None else (None, None)
        call('dig3d_linear_fwd_grouped', G, px, pw, pb, None, M, K, N, act, py, pz, _stream())
        ctx.act, ctx.G, ctx.has_bias = act, G, [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        if zs is None:
            zs = [ys[0].new_empty(0) for _ in range(G)]
            ctx.mark_non_differentiable(*zs)
        ctx.save_for_backward(*xs, *Ws, *zs)
        return tuple(ys) + tuple(zs)

    @staticmethod
    def backward(ctx, *grads):
        G, act = ctx.G, ctx.act
        sv = ctx.saved_tensors
        xs, Ws, zs = sv[:G], sv[G:2 * G], sv[2 * G:3 * G]
        gys, gzs = list(grads[:G]), list(grads[G:2 * G])
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        st = _stream()
        zero = None

        def gy_of(g):
            nonlocal zero
            if gys[g] is not None:
                return _c(gys[g])
            if zero is None:
                zero = torch.zeros(M, N, dtype=torch.float32, device=dev)
            return zero
        if torch.is_grad_enabled():          # create_graph: differentiable input gradients (positions only)
            if any(gz is not None for gz in gzs):
                raise RuntimeError('third-order differentiation through the grouped output blocks is not supported')
            gxs = _GroupedDgradAct.apply(act, G, *[gy_of(g) for g in range(G)], *zs, *Ws)
            return (None, None) + tuple(gxs) + (None,) * (2 * G)
        stride = N * K + N
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        gxs = [torch.empty(M, K, dtype=torch.float32, device=dev) for _ in range(G)]
        kp = [_keyed_partials(Ws[g], nb, stride, stride, dev) for g in range(G)]
        now = kp[0][2]
        use_za = act != ACT_NONE and any(gz is not None for gz in gzs)
        gza = [(_c(gzs[g]) if gzs[g] is not None else None) for g in range(G)] if use_za else None
        if use_za and any(z is None for z in gza):
            zz = torch.zeros(M, N, dtype=torch.float32, device=dev)
            gza = [z if z is not None else zz for z in gza]
        pg, k1 = _ptrs([gy_of(g) for g in range(G)])
        pz, k2 = _ptrs(zs) if act != ACT_NONE else (None, None)
        pw, k3 = _ptrs(Ws)
        px, k4 = _ptrs(xs)
        pgx, k5 = _ptrs(gxs)
        pp, k6 = _ptrs([q[0] for q in kp])
        pgw, k7 = _ptrs([q[1] for q in kp])
        pza, k8 = _ptrs(gza) if gza is not None else (None, None)
        call('dig3d_linear_bwd_grouped', G, pg, pz, pw, px, M, K, N, act, pgx, None, pp, pgw, now, pza, st)
        gws = [(q[1][:N * K].view(N, K) if q[3] else None) for q in kp]
        gbs = [(q[1][N * K:] if hb else None) for q, hb in zip(kp, ctx.has_bias)]
        return (None, None) + tuple(gxs) + tuple(gws) + tuple(gbs)


class _GroupedDgradAct(Function):
    """gx_g = (gy_g * act'(z_g)) W_g for G layers, differentiable w.r.t. (gy, z, W): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, act, G, *tensors):
        gys = [_c(t) for t in tensors[:G]]
        zs = list(tensors[G:2 * G])
        Ws = list(tensors[2 * G:3 * G])
        M, N = gys[0].shape
        K = Ws[0].size(1)
        gxs = [torch.empty(M, K, dtype=torch.float32, device=gys[0].device) for _ in range(G)]
        pg, k1 = _ptrs(gys)
        pz, k2 = _ptrs(zs) if act != ACT_NONE else (None, None)
        pw, k3 = _ptrs(Ws)
        pgx, k4 = _ptrs(gxs)
        call('dig3d_linear_bwd_input_grouped', G, pg, pz, pw, M, K, N, act, pgx, _stream())
        ctx.act, ctx.G = act, G
        ctx.save_for_backward(*gys, *zs, *Ws)
        return tuple(gxs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *ggxs):
        G, act = ctx.G, ctx.act
        sv = ctx.saved_tensors
        gys, zs, Ws = sv[:G], sv[G:2 * G], sv[2 * G:3 * G]
        M, N = gys[0].shape
        K = Ws[0].size(1)
        dev = gys[0].device
        ggxs = [_c(t) for t in ggxs]
        stride = N * K + N
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        kp = [_keyed_partials(Ws[g], nb, stride, N * K, dev) for g in range(G)]
        o_gy = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)]
        o_z = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)] if act != ACT_NONE else None
        pgg, k1 = _ptrs(ggxs)
        pw, k2 = _ptrs(Ws)
        pz, k3 = _ptrs(zs) if act != ACT_NONE else (None, None)
        pgy, k4 = _ptrs(gys)
        pog, k5 = _ptrs(o_gy)
        poz, k6 = _ptrs(o_z) if o_z is not None else (None, None)
        pp, k7 = _ptrs([q[0] for q in kp])
        pgw, k8 = _ptrs([q[1] for q in kp])
        call('dig3d_linear_dd_grouped', G, pgg, pw, pz, pgy, M, K, N, act, pog, poz, pp, pgw, kp[0][2], _stream())
        gws = [(q[1][:N * K].view(N, K) if q[3] else None) for q in kp]
        return (None, None) + tuple(o_gy) + (tuple(o_z) if o_z is not None else (None,) * G) + tuple(gws)


# ---------------------------------------------------------------------------------------------------------------
# The radial projections of the blocks on the energy_and_force route — Y_h = X W_h^T for H bias-free heads over the SAME
# rbf rows (dimenetpp.py:143-145,160: lin_rbf2 lin_rbf1 composed, and lin_rbf) — closed under differentiation on the two
# matrix-core kernels of csrc/radial.hip:
#     F(X, W)  = [X W_h^T]_h             dig3d_radial_fwd
#     A(G, W)  = sum_h G_h W_h           dig3d_radial_bwd (the input gradient, summed over the heads inside the launch)
#     C(G, X)  = [G_h^T X]_h             dig3d_radial_bwd (the weight-gradient partials of the same launch)
# dF = (A, C);  dA/dG_h = F(Q, W)_h, dA/dW_h = C(G, Q)_h.  Forward F; the create_graph backward (the force gradient) A; the
# final backward one A + C launch for the F node and one F + one C launch for the A node — five launches of ~25 us per step
# where the grouped dense kernels at K = 6 took 28 + 47 + 66 + 93 us (k_linear_*_grouped on 128-wide reduction tiles), and
# the 2 L aliases of rbf they needed shrink to one.
# ---------------------------------------------------------------------------------------------------------------
def _rad_tables(Ws, K):
    import ctypes
    H = len(Ws)
    IA = ctypes.c_int * H
    N = [w.size(0) for w in Ws]
    ints = (IA(*N), IA(*N), IA(*([0] * H)), IA(*([0] * H)))
    cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
    pa, keep = _ptr_arr(list(Ws))
    return H, N, ints, cast, pa, keep


def _rad_F_raw(x, Ws):
    M, K = x.shape
    H, N, ints, cast, pa, keep = _rad_tables(Ws, K)
    Y = [torch.empty(M, N[h], dtype=torch.float32, device=x.device) for h in range(H)]
    if M == 0:
        return Y
    py, k2 = _ptr_arr(Y)
    call('dig3d_radial_fwd', ptr(x), M, K, H, pa, None, None, cast(ints[0]), cast(ints[1]), cast(ints[2]), py, _stream())
    return Y


def _rad_AC_raw(x, gYs, Ws, want_gx, want_w):
    """one dig3d_radial_bwd launch: -> (A(G, W) [M, K] or None, [C(G, X)_h [N_h, K] or None]).  Weight gradients of leaf
    weights inside a ``deferred_reductions`` block join the step's one reduction under the weight's key (a weight enters the
    second-order graph twice: the first contribution owns the buffer, the other returns None); the others are reduced here."""
    import ctypes
    from . import ops
    M, K = x.shape
    dev = x.device
    H, N, ints, cast, pa, keep = _rad_tables(Ws, K)
    if M == 0:
        return ((torch.zeros(0, K, dtype=torch.float32, device=dev) if want_gx else None),
                [(torch.zeros(N[h], K, dtype=torch.float32, device=dev) if want_w else None) for h in range(H)])
    stride = _hip.query('dig3d_radial_partial_stride', H, cast(ints[0]), cast(ints[1]), cast(ints[3]), K)
    nb = _hip.query('dig3d_radial_blocks', M, H)
    gX = torch.empty(M, K, dtype=torch.float32, device=dev) if want_gx else None
    work = torch.empty(H * M * K, dtype=torch.float32, device=dev) if (want_gx and H > 1) else None
    part = torch.empty(nb * stride, dtype=torch.float32, device=dev)
    pg, k3 = _ptr_arr([(_c(g) if g is not None else None) for g in gYs])
    call('dig3d_radial_bwd', ptr(x), M, K, H, pa, None, None, cast(ints[0]), cast(ints[1]), cast(ints[2]), pg, ptr(gX), ptr(part),
         ptr(work), _stream())
    gWs = [None] * H
    if want_w:
        d = ops._deferred
        now, off = [], 0
        for h in range(H):
            n = N[h] * K
            view = part[off:]
            if d is not None and Ws[h].is_leaf:
                gwb = d.add_keyed(Ws[h].data_ptr(), view, nb, n, n, dev, row_stride=stride)
                gWs[h] = gwb.view(N[h], K) if gwb is not None else None
            else:
                gwb = torch.empty(n, dtype=torch.float32, device=dev)
                now.append((view, nb, stride, n, gwb))
                gWs[h] = gwb.view(N[h], K)
            off += n + N[h]
        if now:
            ops.deferred_reductions._launch('dig3d_reduce_many', now)
    return gX, gWs


class _RadF(Function):
    @staticmethod
    def forward(ctx, x, *Ws):
        from . import ops
        x = _c(x)
        Ws = [_c(w) for w in Ws]
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, *Ws)
        return tuple(_rad_F_raw(x, Ws))

    @staticmethod
    def backward(ctx, *gYs):
        sv = ctx.saved_tensors
        x, Ws = sv[0], list(sv[1:])
        H = len(Ws)
        if all(g is None for g in gYs):
            return (None,) * (1 + H)
        if torch.is_grad_enabled():              # create_graph: the force gradient — positions only, no weight gradient
            want_w = any(ctx.needs_input_grad[1:])
            if not ctx.pos_only and want_w:
                raise NotImplementedError('dig_amd radial projections: a create_graph backward is supported for the position '
                                          'gradient of an energy_and_force forward only')
            _warn_skipped_wgrad(want_w)
            if not ctx.needs_input_grad[0]:
                return (None,) * (1 + H)
            gs = [(g if g is not None else torch.zeros(x.size(0), Ws[h].size(0), dtype=torch.float32, device=x.device))
                  for h, g in enumerate(gYs)]
            return (_RadA.apply(x.detach(), H, *gs, *Ws),) + (None,) * H
        gX, gWs = _rad_AC_raw(x, gYs, Ws, ctx.needs_input_grad[0], any(ctx.needs_input_grad[1:]))
        return (gX,) + tuple(gWs)


class _RadA(Function):
    """gX = sum_h G_h W_h as a differentiable function of (G, W).  ``xk``: any [M, K] float32 buffer — the kernel's second
    operand (its weight-gradient half is computed and dropped here)."""

    @staticmethod
    def forward(ctx, xk, H, *tensors):
        gs = [_c(t) for t in tensors[:H]]
        Ws = [_c(t) for t in tensors[H:]]
        ctx.H = H
        ctx.save_for_backward(*gs, *Ws)
        return _rad_AC_raw(xk, gs, Ws, True, False)[0]

    @staticmethod
    def backward(ctx, Q):
        H = ctx.H
        sv = ctx.saved_tensors
        gs, Ws = list(sv[:H]), list(sv[H:])
        if Q is None:
            return (None, None) + (None,) * (2 * H)
        if torch.is_grad_enabled():
            raise NotImplementedError('dig_amd radial projections: third-order differentiation is not supported')
        Q = _c(Q)
        dG = _rad_F_raw(Q, Ws) if any(ctx.needs_input_grad[2:2 + H]) else [None] * H
        dW = [None] * H
        if any(ctx.needs_input_grad[2 + H:]):
            dW = _rad_AC_raw(Q, gs, Ws, False, True)[1]
        return (None, None) + tuple(dG) + tuple(dW)


def radial2_supported(x, Ws):
    from . import ops
    return (ops._twice_differentiable and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) > 0
            and 1 <= x.size(1) <= 8 and 1 <= len(Ws) <= 16
            and all(w.dim() == 2 and w.dtype == torch.float32 and w.size(1) == x.size(1) and 8 <= w.size(0) <= 256
                    and w.size(0) % 4 == 0 for w in Ws))


def radial2(x, Ws):
    """[x W_h^T for h] — the bias-free radial projections over the same rows, twice differentiable (see above)."""
    return list(_RadF.apply(x, *Ws))


class _SegSumG(Function):
    """outs[g] = segment_sum(xs[g]) for G tensors over ONE sorted segmentation (csrc/readout.hip:k_segsum_grouped) — with
    ``_GatherG`` a pair closed under differentiation (a linear map and its adjoint), so the energy_and_force step's edge ->
    node sums of all output blocks (spherenet.py:196, dimenetpp.py:176) are one launch in each of its four passes."""

    @staticmethod
    def forward(ctx, seg, G, *xs):
        from . import ops
        xs = [_c(x) for x in xs]
        C = xs[0].size(1)
        outs = [torch.empty(seg.S, C, dtype=torch.float32, device=xs[0].device) for _ in range(G)]
        pi, k1 = ops._ptrs(xs)
        po, k2 = ops._ptrs(outs)
        call('dig3d_segment_sum_grouped', G, pi, None, ptr(seg.kptr), seg.S, C, po, _stream())
        ctx.seg, ctx.G = seg, G
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        return (None, None) + tuple(_GatherG.apply(ctx.seg, ctx.G, *gs))


class _GatherG(Function):
    """outs[g] = xs[g][seg.key] (rows behind the live count of a padded batch: zero): the adjoint of ``_SegSumG``."""

    @staticmethod
    def forward(ctx, seg, G, *xs):
        from . import ops
        xs = [_c(x) for x in xs]
        C, M = xs[0].size(1), seg.key.numel()
        outs = [torch.empty(M, C, dtype=torch.float32, device=xs[0].device) for _ in range(G)]
        pi, k1 = ops._ptrs(xs)
        po, k2 = ops._ptrs(outs)
        call('dig3d_gather_grouped', G, pi, ptr(seg.key), M, C, po, None, None, None, ptr(seg.cnt), _stream())
        ctx.seg, ctx.G = seg, G
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        return (None, None) + tuple(_SegSumG.apply(ctx.seg, ctx.G, *gs))


# ---------------------------------------------------------------------------------------------------------------
# v_g = scatter(r_g * h_g, i) for the G = L + 1 output blocks (dimenetpp.py:160,176: e2 = lin_rbf(rbf) * e1, then the edge ->
# node sum) WITHOUT the products as tensors, closed under differentiation on the two grouped kernels of csrc/readout.hip:
#     F(r, h)      = S(r * h)                          k_segsum_grouped with the product formed while summing
#     B(gv, r, h)  = (G(gv) * h, G(gv) * r)            k_gather_grouped: both factor gradients in one pass
#     dB: d/d gv = S(Qr * h + Qh * r) (two products per group), d/d r = G(gv) * Qh, d/d h = G(gv) * Qr
# (S = segment sum over the sorted targets, G = its adjoint, the row gather).  F hands aliases of its factors to its own
# create_graph backward B; what B's backward sends to them in the final pass arrives at F's backward as arguments and is added
# inside the gather launch (dig3d_gather_grouped_add).  Replaces, per step, 5 k_ew_mul + 10 k_ew_mul_bwd + 4 k_ew_mul_bwd2
# launches of ~6 us and the [E, 128] products they wrote.
# ---------------------------------------------------------------------------------------------------------------
def _msg_segsum(seg, hs, rs, hs2=None, rs2=None):
    from . import ops
    G, C = len(hs), hs[0].size(1)
    outs = [torch.empty(seg.S, C, dtype=torch.float32, device=hs[0].device) for _ in range(G)]
    pi, k1 = ops._ptrs(hs)
    pm, k2 = ops._ptrs(rs)
    po, k3 = ops._ptrs(outs)
    if hs2 is None:
        call('dig3d_segment_sum_grouped', G, pi, pm, ptr(seg.kptr), seg.S, C, po, _stream())
    else:
        pi2, k4 = ops._ptrs(hs2)
        pm2, k5 = ops._ptrs(rs2)
        call('dig3d_segment_sum_grouped2', G, pi, pm, pi2, pm2, ptr(seg.kptr), seg.S, C, po, _stream())
    return outs


def _msg_gather(seg, gvs, m1, m2, add1=None, add2=None):
    """-> ([G(gv_g) * m1_g (+ add1_g)], [G(gv_g) * m2_g (+ add2_g)])"""
    from . import ops
    G, C, M = len(gvs), gvs[0].size(1), seg.key.numel()
    dev = gvs[0].device
    o1 = [torch.empty(M, C, dtype=torch.float32, device=dev) for _ in range(G)]
    o2 = [torch.empty(M, C, dtype=torch.float32, device=dev) for _ in range(G)]
    pi, k1 = ops._ptrs(gvs)
    p1, k2 = ops._ptrs(o1)
    p2, k3 = ops._ptrs(o2)
    pm1, k4 = ops._ptrs(m1)
    pm2, k5 = ops._ptrs(m2)
    pa1, k6 = ops._ptrs(add1) if add1 is not None and any(a is not None for a in add1) else (None, None)
    pa2, k7 = ops._ptrs(add2) if add2 is not None and any(a is not None for a in add2) else (None, None)
    call('dig3d_gather_grouped_add', G, pi, ptr(seg.key), M, C, p1, pm1, p2, pm2, pa1, pa2, ptr(seg.cnt), _stream())
    return o1, o2


class _MulSegSumG(Function):
    @staticmethod
    def forward(ctx, seg, G, *t):
        rs = [_c(x) for x in t[:G]]
        hs = [_c(x) for x in t[G:2 * G]]
        outs = _msg_segsum(seg, hs, rs)
        r2 = [r.view_as(r) for r in rs]
        h2 = [h.view_as(h) for h in hs]
        ctx.seg, ctx.G = seg, G
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*rs, *hs, *r2, *h2)
        return tuple(outs) + tuple(r2) + tuple(h2)

    @staticmethod
    def backward(ctx, *g):
        seg, G = ctx.seg, ctx.G
        sv = ctx.saved_tensors
        rs, hs, r2, h2 = list(sv[:G]), list(sv[G:2 * G]), list(sv[2 * G:3 * G]), list(sv[3 * G:])
        gvs, gr2, gh2 = list(g[:G]), list(g[G:2 * G]), list(g[2 * G:3 * G])
        if all(x is None for x in gvs):
            return (None, None) + tuple(gr2) + tuple(gh2)
        gvs = [(_c(x) if x is not None else torch.zeros(seg.S, rs[0].size(1), dtype=torch.float32, device=rs[0].device))
               for x in gvs]
        if torch.is_grad_enabled():                  # create_graph: differentiable, through the aliases
            outs = _MulGatherG.apply(seg, G, *gvs, *r2, *h2)
            gr = [(a if b is None else a + b) for a, b in zip(outs[:G], gr2)]
            gh = [(a if b is None else a + b) for a, b in zip(outs[G:], gh2)]
            return (None, None) + tuple(gr) + tuple(gh)
        cc = lambda xs: [(_c(x) if x is not None else None) for x in xs]
        gr, gh = _msg_gather(seg, gvs, hs, rs, cc(gr2), cc(gh2))
        return (None, None) + tuple(gr) + tuple(gh)


class _MulGatherG(Function):
    """(gr_g, gh_g) = (G(gv_g) * h_g, G(gv_g) * r_g) as a differentiable function of (gv, r, h)."""

    @staticmethod
    def forward(ctx, seg, G, *t):
        gvs = [_c(x) for x in t[:G]]
        rs = [_c(x) for x in t[G:2 * G]]
        hs = [_c(x) for x in t[2 * G:3 * G]]
        ctx.seg, ctx.G = seg, G
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*gvs, *rs, *hs)
        gr, gh = _msg_gather(seg, gvs, hs, rs)
        return tuple(gr) + tuple(gh)

    @staticmethod
    def backward(ctx, *Q):
        seg, G = ctx.seg, ctx.G
        sv = ctx.saved_tensors
        gvs, rs, hs = list(sv[:G]), list(sv[G:2 * G]), list(sv[2 * G:])
        if all(q is None for q in Q):
            return (None, None) + (None,) * (3 * G)
        if torch.is_grad_enabled():
            raise NotImplementedError('dig_amd product segment sums: third-order differentiation is not supported')
        z = None
        def fill(q):
            nonlocal z
            if q is not None:
                return _c(q)
            if z is None:
                z = torch.zeros_like(rs[0])
            return z
        Qr = [fill(q) for q in Q[:G]]
        Qh = [fill(q) for q in Q[G:]]
        dgv = _msg_segsum(seg, Qr, hs, Qh, rs)              # S(Qr * h + Qh * r)
        dr, dh = _msg_gather(seg, gvs, Qh, Qr)              # G(gv) * Qh, G(gv) * Qr
        return (None, None) + tuple(dgv) + tuple(dr) + tuple(dh)


def mul_segsum_grouped_supported(rs, hs, seg):
    return (len(rs) == len(hs) and segsum_grouped_supported(hs, seg) and all(r.shape == hs[0].shape and r.is_cuda
                                                                              and r.dtype == torch.float32 for r in rs))


def mul_segsum_grouped(rs, hs, seg):
    """[segment_sum(r_g * h_g) for g] over the sorted segmentation ``seg`` — twice differentiable, the products never written."""
    G = len(rs)
    return list(_MulSegSumG.apply(seg, G, *rs, *hs)[:G])


def segsum_grouped_supported(xs, seg):
    return (1 <= len(xs) <= 8 and seg.perm is None and all(x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
                                                           and x.shape == xs[0].shape for x in xs)
            and xs[0].size(1) in (32, 64, 128, 256) and xs[0].size(0) == seg.key.numel() and xs[0].size(0) > 0)


def segsum_grouped(xs, seg):
    return list(_SegSumG.apply(seg, len(xs), *xs))


# ---------------------------------------------------------------------------------------------------------------
# A tensor with n consumers receives n gradients per backward pass and autograd adds them with n - 1 framework launches.
# ``fan_out`` hands every consumer its own alias; the gradients then arrive as separate arguments of ONE backward node, which
# sums them with one launch (fixed order).  Linear, hence closed under differentiation: the backward of the sum hands its
# incoming gradient to every term.  (tools/diag_force_fanin.py: rbf [E, 6] of a DimeNet++ energy_and_force step has 2 + 2 L
# consumers in both backward passes: 18 additions -> 2 launches.)
# ---------------------------------------------------------------------------------------------------------------
class _SumMany(Function):
    @staticmethod
    def forward(ctx, *gs):
        gs = [_c(g) for g in gs]
        out = torch.empty_like(gs[0])
        pp, keep = _ptr_arr(gs)
        call('dig3d_sum_many', pp, len(gs), gs[0].numel(), ptr(out), _stream())
        ctx.n = len(gs)
        return out

    @staticmethod
    def backward(ctx, go):
        return (go,) * ctx.n


class _FanOut(Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        if len(live) > 16 or any(g.shape != live[0].shape or g.dtype != torch.float32 for g in live):
            s = live[0]
            for g in live[1:]:
                s = s + g
            return s, None
        return _SumMany.apply(*live), None


def fan_out(x, n):
    """n aliases of ``x`` whose gradients are summed by one launch (n >= 3; below that autograd's own addition is one launch
    as well)."""
    if n < 3 or n > 16 or not x.is_cuda or x.dtype != torch.float32 or not x.requires_grad:
        return [x] * n
    return list(_FanOut.apply(x, n))


def grouped_linear2(xs, Ws, bs, act):
    """[act(x_g W_g^T + b_g)] for G same-shape layers (N > 64), twice differentiable, one launch per pass."""
    G = len(xs)
    return list(_GroupedLinAct2.apply(act, G, *xs, *Ws, *bs)[:G])


# ---------------------------------------------------------------------------------------------------------------
# The 8-layer residual chain of an interaction block (spherenet.py:172-182), twice differentiable, on the LDS-resident
# chain kernels (csrc/dense.hip): forward k_chain_fwd<false>; the create_graph backward (the force gradient) k_chain_bwd;
# ITS backward k_chain_fwd<true> (same GEMMs and skip pattern as the forward, act' / act'' epilogue) + k_chain_wgrad; the
# final backward k_chain_bwd again (with the act'' terms that reached the pre-activations) + k_chain_wgrad.  Seven
# launches per block instead of 8 layers x 4 passes, and no framework add for any skip connection.
# ---------------------------------------------------------------------------------------------------------------
def _ptr_arr(ts):
    import ctypes
    arr = (ctypes.c_void_p * len(ts))(*[ptr(t) for t in ts])
    return ctypes.cast(arr, ctypes.c_void_p), arr


def _int_arr(vs):
    import ctypes
    arr = (ctypes.c_int * len(vs))(*vs)
    return ctypes.cast(arr, ctypes.c_void_p), arr


def _chain_wgrad(GZ, Xs, Ks, M, weights, n_valid):
    """weight(+bias) gradient buffers of all layers in one launch -> per layer (gwb, mine): ``gwb`` is where the reduction
    lands, ``mine`` whether this contribution hands the WEIGHT part to autograd (see _keyed_partials; the bias part is
    produced by the final backward only and is always returned from there)."""
    from . import ops
    nl = len(GZ)
    dev = GZ[0].device
    d = ops._deferred
    if d is not None and ops.force_wgrad_deferred and all(w.is_leaf for w in weights):
        # the products themselves wait for the pass's one weight-gradient launch (ops.deferred_reductions.add_wgrad: both
        # contributions of a weight under its key, the second one without the bias part)
        return [d.add_wgrad(GZ[l], Xs[l], Ks[l], 128, key=weights[l].data_ptr(), n_valid=n_valid(l)) for l in range(nl)]
    nb = _hip.query('dig3d_chain_wgrad_workers', M, nl)
    rows = [_keyed_partials(weights[l], nb, 128 * Ks[l] + 128, n_valid(l), dev) for l in range(nl)]
    pg, k1 = _ptr_arr(GZ)
    px, k2 = _ptr_arr(Xs)
    pk, k3 = _int_arr(Ks)
    pp, k4 = _ptr_arr([r[0] for r in rows])
    po, k5 = _ptr_arr([r[1] for r in rows])
    call('dig3d_chain_wgrad', nl, pg, px, pk, M, pp, po, rows[0][2], _stream())
    return [(r[1], r[3]) for r in rows]


class _Chain2(Function):
    """(Y_last, Z_0 .. Z_{nl-1}) of the chain; the pre-activations are OUTPUTS so that the second-order pass can send its
    act'' terms back to them."""

    @staticmethod
    def forward(ctx, x0, spec, packed, *tensors):
        from . import ops
        nl = len(spec)
        x0 = _c(x0)
        M = x0.size(0)
        dev = x0.device
        Ws = [_c(tensors[3 * l]) for l in range(nl)]
        bs = [tensors[3 * l + 1] for l in range(nl)]
        rs = [(_c(tensors[3 * l + 2]) if tensors[3 * l + 2] is not None else None) for l in range(nl)]
        Zs = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        Ys = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        pw, k1 = _ptr_arr(Ws)
        pb, k2 = _ptr_arr(bs)
        pr, k3 = _ptr_arr(rs)
        pz, k4 = _ptr_arr(Zs)
        py, k5 = _ptr_arr(Ys)
        pk, k6 = _int_arr([sp[0] for sp in spec])
        pres, k7 = _int_arr([sp[2] for sp in spec])
        psv, k8 = _int_arr([sp[3] for sp in spec])
        pa, k9 = _int_arr([sp[1] for sp in spec])
        # first-order passes (this forward, the create_graph backward, the final backward) run on the packed-weight
        # kernels of csrc/chain.hip (47 / 52 us per chain at E ~ 8.7k instead of 87 / 86); only the second-order pass
        # (k_chain_fwd<true>) still reads the row-major weights.  ``packed``: this chain's slice of the model's one pack launch
        if packed is None:
            packed = ops.pack_weights(Ws)
        call('dig3d_chainp_fwd', ptr(x0), M, nl, ptr(packed[0]), pb, pr, pz, py, pk, pres, psv, pa, _stream())
        ctx.spec = spec
        ctx.has_bias = [b is not None for b in bs]
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x0, *Ws, *Zs, *Ys[:-1], packed)
        return (Ys[-1],) + tuple(Zs)

    @staticmethod
    def backward(ctx, gy, *gzs):
        spec = ctx.spec
        nl = len(spec)
        sv = ctx.saved_tensors
        x0, Ws, Zs, Ys, packed = sv[0], sv[1:1 + nl], sv[1 + nl:1 + 2 * nl], sv[1 + 2 * nl:-1], sv[-1]
        M = x0.size(0)
        dev = x0.device
        none = (None, None, None) + (None,) * (3 * nl)
        if gy is None and all(g is None for g in gzs):
            return none
        Ks = [sp[0] for sp in spec]
        ext = [l for l in range(nl) if spec[l][2] == 1]
        if gy is None:
            gy = torch.zeros(M, 128, dtype=torch.float32, device=dev)
        if torch.is_grad_enabled():          # create_graph=True: the force gradient, itself differentiable
            if not ctx.pos_only or any(g is not None for g in gzs):
                raise NotImplementedError('dig_amd chain: a create_graph backward is supported for the position gradient '
                                          'of an energy_and_force forward only')
            outs = _ChainBwd2.apply(gy, spec, packed, *Ws, *Zs)
            grads = [None] * (3 * nl)
            for k, l in enumerate(ext):
                grads[3 * l + 2] = outs[1 + k]
            return (outs[0], None, None) + tuple(grads)
        # final backward: input gradient recursion (with the act'' terms that reached the pre-activations) + weights
        GZ = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        gres = [torch.empty(M, 128, dtype=torch.float32, device=dev) if l in ext else None for l in range(nl)]
        gx0 = torch.empty(M, Ks[0], dtype=torch.float32, device=dev)
        pw, k1 = _ptr_arr(Ws)
        pz, k2 = _ptr_arr(Zs)
        pg, k3 = _ptr_arr(GZ)
        pr, k4 = _ptr_arr(gres)
        pa, k5 = _ptr_arr([(_c(g) if g is not None else None) for g in gzs])
        pk, k6 = _int_arr(Ks)
        pres, k7 = _int_arr([sp[2] for sp in spec])
        psv, k8 = _int_arr([sp[3] for sp in spec])
        pact, k9 = _int_arr([sp[1] for sp in spec])
        call('dig3d_chainp_bwd', ptr(_c(gy)), M, nl, ptr(packed[1]), pz, pg, pr, pk, pres, psv, pact, ptr(gx0), None, pa,
             _stream())
        gwbs = _chain_wgrad(GZ, [x0] + list(Ys), Ks, M, Ws, lambda l: 128 * Ks[l] + 128)
        grads = []
        for l in range(nl):
            (gwb, mine), K = gwbs[l], Ks[l]
            grads += [gwb[:128 * K].view(128, K) if mine else None, gwb[128 * K:] if ctx.has_bias[l] else None, gres[l]]
        return (gx0, None, None) + tuple(grads)


class _ChainBwd2(Function):
    """(gx0, gres of the external residuals...) = the input-gradient recursion of the chain as a differentiable function
    of (gout, W_l, Z_l)."""

    @staticmethod
    def forward(ctx, gout, spec, packed, *tensors):
        nl = len(spec)
        gout = _c(gout)
        Ws, Zs = tensors[:nl], tensors[nl:2 * nl]
        M = gout.size(0)
        dev = gout.device
        Ks = [sp[0] for sp in spec]
        ext = [l for l in range(nl) if spec[l][2] == 1]
        GZ = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        G = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        gres = [torch.empty(M, 128, dtype=torch.float32, device=dev) if l in ext else None for l in range(nl)]
        gx0 = torch.empty(M, Ks[0], dtype=torch.float32, device=dev)
        pw, k1 = _ptr_arr(Ws)
        pz, k2 = _ptr_arr(Zs)
        pg, k3 = _ptr_arr(GZ)
        pr, k4 = _ptr_arr(gres)
        pG, k5 = _ptr_arr(G)
        pk, k6 = _int_arr(Ks)
        pres, k7 = _int_arr([sp[2] for sp in spec])
        psv, k8 = _int_arr([sp[3] for sp in spec])
        pact, k9 = _int_arr([sp[1] for sp in spec])
        call('dig3d_chainp_bwd', ptr(gout), M, nl, ptr(packed[1]), pz, pg, pr, pk, pres, psv, pact, ptr(gx0), pG, None,
             _stream())
        ctx.spec = spec
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*Ws, *Zs, *GZ, *G, packed)
        return (gx0,) + tuple(gres[l] for l in ext)

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx0, *ggres):
        spec = ctx.spec
        nl = len(spec)
        sv = ctx.saved_tensors
        Ws, Zs, GZ, G, packed = sv[:nl], sv[nl:2 * nl], sv[2 * nl:3 * nl], sv[3 * nl:4 * nl], sv[4 * nl]
        M = GZ[0].size(0)
        dev = GZ[0].device
        Ks = [sp[0] for sp in spec]
        ext = [l for l in range(nl) if spec[l][2] == 1]
        if ggx0 is None and all(g is None for g in ggres):
            return (None, None, None) + (None,) * (2 * nl)
        ggx0 = _c(ggx0) if ggx0 is not None else torch.zeros(M, Ks[0], dtype=torch.float32, device=dev)
        rr = [None] * nl
        for k, l in enumerate(ext):
            rr[l] = _c(ggres[k]) if ggres[k] is not None else None
        HZ = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        U = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        pw, k1 = _ptr_arr(Ws)
        pz, k2 = _ptr_arr(Zs)
        pG, k3 = _ptr_arr(G)
        pr, k4 = _ptr_arr(rr)
        ph, k5 = _ptr_arr(HZ)
        pu, k6 = _ptr_arr(U)
        pk, k7 = _int_arr(Ks)
        pres, k8 = _int_arr([sp[2] for sp in spec])
        psv, k9 = _int_arr([sp[3] for sp in spec])
        pact, k10 = _int_arr([sp[1] for sp in spec])
        if _OLD_CHAIN_DD:     # the round-2 LDS kernel on the row-major weights (tests compare the two)
            call('dig3d_chain_dd', ptr(ggx0), M, nl, pw, pz, pG, pr, ph, pu, pk, pres, psv, pact, _stream())
        else:                 # the register-resident kernel on the weights packed by the forward (csrc/chain.hip)
            call('dig3d_chainp_dd', ptr(ggx0), M, nl, ptr(packed[0]), pz, pG, pr, ph, pu, pk, pres, psv, pact, _stream())
        gwbs = _chain_wgrad(list(GZ), [ggx0] + U[:-1], Ks, M, Ws, lambda l: 128 * Ks[l])
        gws = [(gwbs[l][0][:128 * Ks[l]].view(128, Ks[l]) if gwbs[l][1] else None) for l in range(nl)]
        return (U[-1], None, None) + tuple(gws) + tuple(HZ)


_OLD_CHAIN_DD = False   # True: the second-order pass of chain2 on the round-2 kernel (dig3d_chain_dd)
_NO_CHAIN2 = False      # True: per-layer twice-differentiable Functions instead of chain2 (tests compare the two)


def chain2_supported(x0, layers):
    """the twice-differentiable chain: <= 8 swish layers of 128 outputs, K_0 <= 128 (multiple of 8), K_l = 128 afterwards."""
    from . import ops
    if _NO_CHAIN2 or not ops._twice_differentiable or not (1 <= len(layers) <= 8) or not x0.is_cuda:
        return False
    if x0.dim() != 2 or x0.size(0) == 0 or x0.dtype != torch.float32:
        return False
    for l, (w, b, act, res, rt, save) in enumerate(layers):
        K = w.size(1)
        if w.size(0) != 128 or K > 128 or K % 8 or (l > 0 and K != 128) or act != ACT_SWISH or not w.is_leaf:
            return False
    return layers[0][0].size(1) == x0.size(1)


def chain2(x0, layers, packed=None):
    spec = tuple((w.size(1), act, res, int(bool(save))) for (w, b, act, res, rt, save) in layers)
    flat = []
    for (w, b, act, res, rt, save) in layers:
        flat += [w, b, rt if res == 1 else None]
    return _Chain2.apply(x0, spec, packed, *flat)[0]


# ---------------------------------------------------------------------------------------------------------------
# The FRONT of an interaction block (dimenetpp.py:130-145, spherenet.py:150-163), twice differentiable on three launches
# (csrc/chain.hip): forward = the energy route's dig3d_front_fwd; the create_graph backward (the force gradient) = k_front_bwd
# (keeping the gradient that reached the product); ITS backward = k_front_dd (the forward's three products with act' / act''
# epilogues) + one weight-gradient launch; the final backward = k_front_bwd again, with the act'' terms that reached the three
# pre-activations added in its epilogues, + one weight-gradient launch.  Replaces per block and step: a grouped
# twice-differentiable pair (lin_ji, lin_kj), the product with the radial projection (diffops.mul2) and lin_down as separate
# Functions — 3 launches + 2-6 framework additions on [E, 128] in each of the four passes (r05: 412 launches per config-3 step).
# ---------------------------------------------------------------------------------------------------------------
def _front_wgrad(GZs, Xs, Ns, M, weights, n_valid):
    """weight(+bias) gradient buffers of (lin_ji, lin_kj, lin_down) in one launch -> [(gwb, mine)] (see _chain_wgrad)."""
    from . import ops
    dev = GZs[0].device
    d = ops._deferred
    if d is not None and ops.force_wgrad_deferred and all(w.is_leaf for w in weights):
        return [d.add_wgrad(GZs[l], Xs[l], 128, Ns[l], key=weights[l].data_ptr(), n_valid=n_valid(l)) for l in range(3)]
    nb = _hip.query('dig3d_chain_wgrad_workers', M, 3)
    rows = [_keyed_partials(weights[l], nb, Ns[l] * 128 + Ns[l], n_valid(l), dev) for l in range(3)]
    pg, k1 = _ptr_arr(GZs)
    px, k2 = _ptr_arr(Xs)
    pk, k3 = _int_arr([128, 128, 128])
    pn, k4 = _int_arr(list(Ns))
    pp, k5 = _ptr_arr([r[0] for r in rows])
    po, k6 = _ptr_arr([r[1] for r in rows])
    call('dig3d_chain_wgrad_n', 3, pg, px, pk, pn, M, pp, po, rows[0][2], _stream())
    return [(r[1], r[3]) for r in rows]


class _Front2(Function):
    """(x_ji, xd, Zji, Zkj, Zd) = front(x1, rb): the pre-activations are OUTPUTS so that the second-order pass can send its
    act'' terms back to them (as _Chain2 does)."""

    @staticmethod
    def forward(ctx, x1, rb, Wji, bji, Wkj, bkj, Wd, packed=None):
        from . import ops
        x1, rb = _c(x1), _c(rb)
        M, ND = x1.size(0), Wd.size(0)
        dev = x1.device
        if packed is None:                 # else: this front's slice of the model's one pack launch
            packed = ops.pack_weights([Wji, Wkj, Wd])
        Zji, Xji, Zkj, T = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(4))
        Zd, Xd = (torch.empty(M, ND, dtype=torch.float32, device=dev) for _ in range(2))
        call('dig3d_front_fwd', ptr(x1), M, ptr(packed[0]), ptr(bji), ptr(bkj), ptr(rb), ptr(Zji), ptr(Xji), ptr(Zkj),
             ptr(T), ptr(Zd), ptr(Xd), ND, _stream())
        ctx.ND = ND
        ctx.has_bias = (bji is not None, bkj is not None)
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        rb2 = rb.view_as(rb)          # alias of rb for this front's own create_graph backward (its gradient comes back below)
        ctx.save_for_backward(x1, rb, Zji, Zkj, Zd, T, Wji, Wkj, Wd, packed, rb2)
        # the last two outputs are aliases of x1 for its OTHER consumers (the skip connection of the layer chain, the e2 product
        # of the previous block): their gradients come back as separate arguments and are added inside k_front_bwd instead of
        # by two framework additions per block and pass (as ops._Front does on the energy route)
        return Xji, Xd, Zji, Zkj, Zd, x1.view_as(x1), x1.view_as(x1), rb2

    @staticmethod
    def backward(ctx, gxji, gxd, gzji, gzkj, gzd, ga0, ga1, grb2):
        x1, rb, Zji, Zkj, Zd, T, Wji, Wkj, Wd, packed, rb2 = ctx.saved_tensors
        M, ND = x1.size(0), ctx.ND
        dev = x1.device
        if all(g is None for g in (gxji, gxd, gzji, gzkj, gzd, ga0, ga1, grb2)):
            return (None,) * 8
        ga0 = _c(ga0) if ga0 is not None else None
        ga1 = _c(ga1) if ga1 is not None else None
        gxji = _c(gxji) if gxji is not None else torch.zeros(M, 128, dtype=torch.float32, device=dev)
        gxd = _c(gxd) if gxd is not None else torch.zeros(M, ND, dtype=torch.float32, device=dev)
        if torch.is_grad_enabled():          # create_graph=True: the force gradient, itself differentiable
            if not ctx.pos_only or any(g is not None for g in (gzji, gzkj, gzd)):
                raise NotImplementedError('dig_amd front: a create_graph backward is supported for the position gradient of '
                                          'an energy_and_force forward only')
            gx1, grb = _FrontBwd2.apply(gxji, gxd, rb2, Zji, Zkj, Zd, Wji, Wkj, Wd, packed, ga0, ga1)
            return gx1, (grb if grb2 is None else grb + grb2), None, None, None, None, None, None
        GZji, GZkj, grb, gx1 = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(4))
        GZd = torch.empty(M, ND, dtype=torch.float32, device=dev)
        opt = lambda g: ptr(_c(g)) if g is not None else None
        # (what this front's own create_graph backward sent to rb' is added to grb inside the launch)
        call('dig3d_front_bwd_add', M, ptr(packed[1]), ptr(Zd), ptr(Zkj), ptr(Zji), ptr(rb), ptr(gxd), ptr(gxji), ptr(ga0), ptr(ga1),
             ptr(GZd), ptr(GZkj), ptr(GZji), ptr(grb), ptr(gx1), ND, None, opt(gzd), opt(gzkj), opt(gzji), opt(grb2), _stream())
        Ns = (128, 128, ND)
        gw = _front_wgrad([GZji, GZkj, GZd], [x1, x1, T], Ns, M, [Wji, Wkj, Wd], lambda l: Ns[l] * 128 + Ns[l])
        gW = [(gw[l][0][:Ns[l] * 128].view(Ns[l], 128) if gw[l][1] else None) for l in range(3)]
        gb = [gw[l][0][Ns[l] * 128:] for l in range(2)]
        return (gx1, grb, gW[0], gb[0] if ctx.has_bias[0] else None, gW[1], gb[1] if ctx.has_bias[1] else None, gW[2], None)


class _FrontBwd2(Function):
    """(gx1, grb) = the input gradients of the front as a differentiable function of (gxji, gxd, rb, Z*, W*)."""

    @staticmethod
    def forward(ctx, gxji, gxd, rb, Zji, Zkj, Zd, Wji, Wkj, Wd, packed, ga0=None, ga1=None):
        gxji, gxd = _c(gxji), _c(gxd)
        M, ND = gxji.size(0), gxd.size(1)
        dev = gxji.device
        GZji, GZkj, grb, gx1, Gm = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(5))
        GZd = torch.empty(M, ND, dtype=torch.float32, device=dev)
        # gx1 = (front's own input gradient) + ga0 + ga1: linear in the two extra gradients, their derivative is the identity
        call('dig3d_front_bwd', M, ptr(packed[1]), ptr(Zd), ptr(Zkj), ptr(Zji), ptr(rb), ptr(gxd), ptr(gxji), ptr(ga0), ptr(ga1),
             ptr(GZd), ptr(GZkj), ptr(GZji), ptr(grb), ptr(gx1), ND, ptr(Gm), None, None, None, _stream())
        ctx.ND = ND
        ctx.has_ga = (ga0 is not None, ga1 is not None)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(gxji, gxd, rb, Zji, Zkj, Zd, Wji, Wkj, Wd, packed, GZji, GZkj, GZd, Gm)
        return gx1, grb

    @staticmethod
    @once_differentiable
    def backward(ctx, U, V):
        gxji, gxd, rb, Zji, Zkj, Zd, Wji, Wkj, Wd, packed, GZji, GZkj, GZd, Gm = ctx.saved_tensors
        M, ND = gxji.size(0), ctx.ND
        dev = gxji.device
        if U is None and V is None:
            return (None,) * 12
        U = _c(U) if U is not None else torch.zeros(M, 128, dtype=torch.float32, device=dev)
        V = _c(V) if V is not None else None
        dgxji, HZji, HZkj, drb, Cgm = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(5))
        dgxd, HZd = (torch.empty(M, ND, dtype=torch.float32, device=dev) for _ in range(2))
        call('dig3d_front_dd', ptr(U), ptr(V), M, ptr(packed[0]), ptr(Zji), ptr(Zkj), ptr(Zd), ptr(rb), ptr(gxji), ptr(gxd),
             ptr(Gm), ptr(dgxji), ptr(HZji), ptr(HZkj), ptr(drb), ptr(Cgm), ptr(dgxd), ptr(HZd), ND, _stream())
        Ns = (128, 128, ND)
        gw = _front_wgrad([GZji, GZkj, GZd], [U, U, Cgm], Ns, M, [Wji, Wkj, Wd], lambda l: Ns[l] * 128)
        gW = [(gw[l][0][:Ns[l] * 128].view(Ns[l], 128) if gw[l][1] else None) for l in range(3)]
        return (dgxji, dgxd, drb, HZji, HZkj, HZd, gW[0], gW[1], gW[2], None,
                U if ctx.has_ga[0] else None, U if ctx.has_ga[1] else None)


def front2_supported(x1, rb, lin_ji, lin_kj, lin_down):
    from . import ops
    nd = lin_down.out_features
    return (ops._twice_differentiable and x1.is_cuda and x1.dtype == torch.float32 and x1.dim() == 2 and x1.size(0) > 0
            and x1.size(1) == 128 and rb.shape == x1.shape and rb.dtype == torch.float32
            and lin_ji.weight.shape == (128, 128) and lin_kj.weight.shape == (128, 128) and lin_ji.weight.is_leaf
            and lin_kj.weight.is_leaf and lin_down.weight.is_leaf
            and lin_down.in_features == 128 and lin_down.bias is None and nd % 16 == 0 and 16 <= nd <= 128)


def front2(x1, rb, lin_ji, lin_kj, lin_down, packed=None):
    """-> (x_ji, xd, x1', x1''): swish(lin_ji(x1)), swish(lin_down(swish(lin_kj(x1)) * rb)) and two aliases of x1 for its other
    consumers (see ``_Front2``); twice differentiable, one launch per pass."""
    out = _Front2.apply(x1, rb, lin_ji.weight, lin_ji.bias, lin_kj.weight, lin_kj.bias, lin_down.weight, packed)
    return out[0], out[1], out[5], out[6]


# ---------------------------------------------------------------------------------------------------------------
# G independent chains of <= 4 layers with 256 outputs (csrc/wide.hip) — the output blocks of all interaction layers
# (dimenetpp.py:164-195: lin_up 128 -> 256, then lins 256 -> 256 with swish), twice differentiable on ONE launch per pass:
# forward k_wide_fwd; the create_graph backward k_wide_bwd (keeping the total gradient of every layer output); ITS backward
# k_wide_fwd<true> (the forward's products, act' / act'' epilogues); the final backward k_wide_bwd with the act'' terms added
# to the pre-activation gradients.  Weight gradients: the pass's one deferred launch (ops.deferred_reductions.add_wgrad).
# Replaces four grouped launches per pass (diffops.grouped_linear2: 16 of the step's launches, 18-40 us each at 672 rows).
# ---------------------------------------------------------------------------------------------------------------
def _wide_wgrads(GZ, X, Ks, weights, with_bias):
    """-> [(gwb, mine)] for n layers [256, K]: deferred into the pass's weight-gradient launch when one is open."""
    from . import ops
    n = len(GZ)
    d = ops._deferred
    if d is not None and all(w.is_leaf for w in weights):
        return [d.add_wgrad(GZ[i], X[i], Ks[i], 256, key=weights[i].data_ptr(),
                            n_valid=None if with_bias else 256 * Ks[i]) for i in range(n)]
    with ops.deferred_reductions() as red:
        gwbs = [red.add_wgrad(GZ[i], X[i], Ks[i], 256) for i in range(n)]
    red.flush()
    return [(g, True) for g in gwbs]


class _Wide2(Function):
    """tensors = xs[G], then per group and layer (weight, bias).  -> (Y_last of every group ..., Z of every group and activated
    layer ...): the pre-activations are OUTPUTS (see _Chain2)."""

    @staticmethod
    def forward(ctx, G, spec, *tensors):
        import ctypes
        from . import ops
        nl = len(spec)
        xs = [_c(t) for t in tensors[:G]]
        rest = tensors[G:]
        n = G * nl
        Ws = [_c(rest[2 * i]) for i in range(n)]
        bs = [rest[2 * i + 1] for i in range(n)]
        M, K0 = xs[0].shape
        dev = xs[0].device
        Ks = [K0 if (i % nl) == 0 else 256 for i in range(n)]
        packed = torch.empty(n, 2, 65536, dtype=torch.float32, device=dev)
        pw, k1 = _ptr_arr(Ws)
        pk, k2 = _int_arr(Ks)
        pf, k3 = _ptr_arr([packed[i, 0] for i in range(n)])
        pb_, k4 = _ptr_arr([packed[i, 1] for i in range(n)])
        call('dig3d_wide_pack', n, pw, pk, pf, pb_, _stream())
        acts = [l for l in range(nl) if spec[l][0] != ACT_NONE]
        Zs = [torch.empty(M, 256, dtype=torch.float32, device=dev) if (i % nl) in acts else None for i in range(n)]
        Ys = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        px, k5 = _ptr_arr(xs)
        pbias, k6 = _ptr_arr(bs)
        pz, k7 = _ptr_arr(Zs)
        py, k8 = _ptr_arr(Ys)
        pa, k9 = _int_arr([1 if sp[0] == ACT_SWISH else 0 for sp in spec])
        pr, k10 = _int_arr([int(sp[1]) for sp in spec])
        call('dig3d_wide_fwd', G, nl, M, K0, px, pf, pbias, pz, py, pa, pr, _stream())
        ctx.G, ctx.spec, ctx.K0, ctx.acts = G, spec, K0, acts
        ctx.has_bias = [b is not None for b in bs]
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        zs_out = [Zs[g * nl + l] for g in range(G) for l in acts]
        ctx.save_for_backward(packed, *xs, *Ws, *zs_out, *[Ys[i] for i in range(n) if (i % nl) != nl - 1])
        return tuple(Ys[g * nl + nl - 1] for g in range(G)) + tuple(zs_out)

    @staticmethod
    def backward(ctx, *grads):
        G, spec, K0, acts = ctx.G, ctx.spec, ctx.K0, ctx.acts
        nl, na = len(spec), len(acts)
        n = G * nl
        sv = ctx.saved_tensors
        packed, xs, Ws = sv[0], sv[1:1 + G], sv[1 + G:1 + G + n]
        zs_out = sv[1 + G + n:1 + G + n + G * na]
        inner = list(sv[1 + G + n + G * na:])
        M = xs[0].size(0)
        dev = xs[0].device
        gys, gzs = list(grads[:G]), list(grads[G:])
        none = (None, None) + (None,) * (G + 2 * n)
        if all(g is None for g in gys) and all(g is None for g in gzs):
            return none
        gys = [_c(g) if g is not None else torch.zeros(M, 256, dtype=torch.float32, device=dev) for g in gys]
        if torch.is_grad_enabled():          # create_graph=True: the force gradient
            if not ctx.pos_only or any(g is not None for g in gzs):
                raise NotImplementedError('dig_amd wide chain: a create_graph backward is supported for the position gradient '
                                          'of an energy_and_force forward only')
            gx0 = _WideBwd2.apply(G, spec, K0, packed, *gys, *zs_out, *Ws)
            return (None, None) + tuple(gx0) + (None,) * (2 * n)
        Zfull = [None] * n
        gzadd = [None] * n
        for g in range(G):
            for k, l in enumerate(acts):
                Zfull[g * nl + l] = zs_out[g * na + k]
                gz = gzs[g * na + k]
                gzadd[g * nl + l] = _c(gz) if gz is not None else None
        Xin, it = [], iter(inner)
        for g in range(G):
            Xin.append(xs[g])
            for l in range(nl - 1):
                Xin.append(next(it))
        GZ = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        gx0 = [torch.empty(M, K0, dtype=torch.float32, device=dev) for _ in range(G)]
        pg, k1 = _ptr_arr(gys)
        pw, k2 = _ptr_arr([packed[i, 1] for i in range(n)])
        pz, k3 = _ptr_arr(Zfull)
        pgz, k4 = _ptr_arr(GZ)
        pgx, k5 = _ptr_arr(gx0)
        pa, k6 = _int_arr([1 if sp[0] == ACT_SWISH else 0 for sp in spec])
        pr, k7 = _int_arr([int(sp[1]) for sp in spec])
        pza, k8 = _ptr_arr(gzadd)
        call('dig3d_wide_bwd', G, nl, M, K0, pg, pw, pz, pgz, pgx, None, pa, pr, None, pza, _stream())
        Ks = [K0 if (i % nl) == 0 else 256 for i in range(n)]
        gw = _wide_wgrads(GZ, Xin, Ks, Ws, True)
        out = []
        for i in range(n):
            gwb, mine = gw[i]
            out += [gwb[:256 * Ks[i]].view(256, Ks[i]) if mine else None, gwb[256 * Ks[i]:] if ctx.has_bias[i] else None]
        return (None, None) + tuple(gx0) + tuple(out)


class _WideBwd2(Function):
    """gx0[G] = the input gradients of the G chains as a differentiable function of (gout_g, Z, W)."""

    @staticmethod
    def forward(ctx, G, spec, K0, packed, *tensors):
        nl = len(spec)
        n = G * nl
        acts = [l for l in range(nl) if spec[l][0] != ACT_NONE]
        na = len(acts)
        gys = [_c(t) for t in tensors[:G]]
        zs_out = tensors[G:G + G * na]
        Ws = tensors[G + G * na:]
        M = gys[0].size(0)
        dev = gys[0].device
        Zfull = [None] * n
        for g in range(G):
            for k, l in enumerate(acts):
                Zfull[g * nl + l] = zs_out[g * na + k]
        GZ = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        Gt = [torch.empty(M, 256, dtype=torch.float32, device=dev) if (i % nl) in acts else None for i in range(n)]
        gx0 = [torch.empty(M, K0, dtype=torch.float32, device=dev) for _ in range(G)]
        pg, k1 = _ptr_arr(gys)
        pw, k2 = _ptr_arr([packed[i, 1] for i in range(n)])
        pz, k3 = _ptr_arr(Zfull)
        pgz, k4 = _ptr_arr(GZ)
        pgx, k5 = _ptr_arr(gx0)
        pa, k6 = _int_arr([1 if sp[0] == ACT_SWISH else 0 for sp in spec])
        pr, k7 = _int_arr([int(sp[1]) for sp in spec])
        pG, k8 = _ptr_arr(Gt)
        call('dig3d_wide_bwd', G, nl, M, K0, pg, pw, pz, pgz, pgx, None, pa, pr, pG, None, _stream())
        ctx.G, ctx.spec, ctx.K0, ctx.acts = G, spec, K0, acts
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(packed, *zs_out, *Ws, *GZ, *[Gt[g * nl + l] for g in range(G) for l in acts])
        return tuple(gx0)

    @staticmethod
    @once_differentiable
    def backward(ctx, *H0):
        G, spec, K0, acts = ctx.G, ctx.spec, ctx.K0, ctx.acts
        nl, na = len(spec), len(acts)
        n = G * nl
        sv = ctx.saved_tensors
        packed = sv[0]
        zs_out = sv[1:1 + G * na]
        Ws = sv[1 + G * na:1 + G * na + n]
        GZ = sv[1 + G * na + n:1 + G * na + 2 * n]
        Gts = sv[1 + G * na + 2 * n:]
        M = GZ[0].size(0)
        dev = GZ[0].device
        none = (None,) * (4 + G + G * na + n)
        if all(h is None for h in H0):
            return none
        H0 = [_c(h) if h is not None else torch.zeros(M, K0, dtype=torch.float32, device=dev) for h in H0]
        Z0, G0 = [None] * n, [None] * n
        for g in range(G):
            for k, l in enumerate(acts):
                Z0[g * nl + l], G0[g * nl + l] = zs_out[g * na + k], Gts[g * na + k]
        U = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        HZ = [torch.empty(M, 256, dtype=torch.float32, device=dev) if (i % nl) in acts else None for i in range(n)]
        ph, k1 = _ptr_arr(H0)
        pw, k2 = _ptr_arr([packed[i, 0] for i in range(n)])
        pz, k3 = _ptr_arr(Z0)
        pg, k4 = _ptr_arr(G0)
        phz, k5 = _ptr_arr(HZ)
        pu, k6 = _ptr_arr(U)
        pa, k7 = _int_arr([1 if sp[0] == ACT_SWISH else 0 for sp in spec])
        pr, k8 = _int_arr([int(sp[1]) for sp in spec])
        call('dig3d_wide_dd', G, nl, M, K0, ph, pw, pz, pg, phz, pu, pa, pr, _stream())
        Ks = [K0 if (i % nl) == 0 else 256 for i in range(n)]
        Xin = [H0[i // nl] if (i % nl) == 0 else U[i - 1] for i in range(n)]
        gw = _wide_wgrads(list(GZ), Xin, Ks, Ws, False)
        gW = [(gw[i][0][:256 * Ks[i]].view(256, Ks[i]) if gw[i][1] else None) for i in range(n)]
        return ((None,) * 4 + tuple(U[g * nl + nl - 1] for g in range(G))
                + tuple(HZ[g * nl + l] for g in range(G) for l in acts) + tuple(gW))


def wide2_supported(xs, layers):
    """xs: G inputs of one shape [M, 128 | 256]; layers: per group a list of (weight [256, K], bias, act, res) — the shapes of
    csrc/wide.hip, leaf weights, inside an energy_and_force forward."""
    from . import ops
    if not ops._twice_differentiable or not xs or len(xs) > 8 or len(layers) != len(xs):
        return False
    nl = len(layers[0])
    M, K0 = xs[0].shape if xs[0].dim() == 2 else (0, 0)
    if not (1 <= nl <= 4) or K0 not in (128, 256) or M == 0:
        return False
    for x, ls in zip(xs, layers):
        if not x.is_cuda or x.dtype != torch.float32 or tuple(x.shape) != (M, K0) or len(ls) != nl:
            return False
        for l, (w, b, act, res) in enumerate(ls):
            if tuple(w.shape) != (256, K0 if l == 0 else 256) or act not in (ACT_NONE, ACT_SWISH) or not w.is_leaf:
                return False
            if (act, bool(res)) != (layers[0][l][2], bool(layers[0][l][3])) or (res and w.size(1) != 256):
                return False
    return bool(_hip.query('dig3d_wide_supported', M, K0, nl, len(xs)))


def wide2(xs, layers):
    """-> list of the G chain outputs [M, 256], twice differentiable (see ``_Wide2``)."""
    spec = tuple((act, int(bool(res))) for (_, _, act, res) in layers[0])
    flat = []
    for ls in layers:
        for (w, b, _, _) in ls:
            flat += [w, b]
    return list(_Wide2.apply(len(xs), spec, *xs, *flat)[:len(xs)])


# ---------------------------------------------------------------------------------------------------------------
# the 256 -> out_channels heads of the output blocks (spherenet.py:216 ``self.lin(v)``, bias-free), all blocks in one
# launch, twice differentiable on the row-dot kernels of csrc/readout.hip:
#     y = v W^T                      k_smalln_fwd_grouped
#     gv = gy W  (and gW = gy^T v)   k_smalln_bwd_grouped
# are closed under differentiation: d(gy W)/d(gy) . ggv = ggv W^T (the forward kernel), d/dW = gy^T ggv (the weight part of
# the backward kernel).  Replaces 25 GEMV-shaped hipBLASLt launches per energy_and_force step.
# ---------------------------------------------------------------------------------------------------------------
def _smalln_bwd(gys, Ws, Xs, want_gx, weights):
    """-> (gxs or None, [(gwb, mine)] or None)."""
    G = len(gys)
    M, N = gys[0].shape
    K = Ws[0].size(1)
    dev = gys[0].device
    stride = N * K + N
    nb = _hip.query('dig3d_smalln_blocks', M)
    gxs = [torch.empty(M, K, dtype=torch.float32, device=dev) for _ in range(G)] if want_gx else [None] * G
    if weights is not None:
        rows = [_keyed_partials(weights[g], nb, stride, N * K, dev) for g in range(G)]
        parts = [r[0] for r in rows]
    else:
        rows = None
        parts = [None] * G             # no weight partials wanted: the kernel skips that half (and needs no X operand)
    pg, k1 = _ptr_arr(gys)
    pw, k2 = _ptr_arr(Ws)
    px, k3 = _ptr_arr(Xs)
    pgx, k4 = _ptr_arr(gxs)
    pp, k5 = _ptr_arr(parts)
    call('dig3d_smalln_bwd_grouped', G, pg, pw, px, M, K, N, pgx, pp, _stream())
    if rows is not None and rows[0][2]:
        import ctypes
        IA, LA = ctypes.c_int * G, ctypes.c_int64 * G
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        po, k6 = _ptr_arr([r[1] for r in rows])
        call('dig3d_reduce_many', pp, cast(IA(*[nb] * G)), cast(LA(*[stride] * G)), cast(IA(*[stride] * G)), po, G, _stream())
    return (gxs if want_gx else None), ([(r[1], r[3]) for r in rows] if rows is not None else None)


class _Heads2(Function):
    @staticmethod
    def forward(ctx, G, *tensors):
        from . import ops
        vs = [_c(t) for t in tensors[:G]]
        Ws = [_c(t) for t in tensors[G:2 * G]]
        M, K = vs[0].shape
        N = Ws[0].size(0)
        ys = [torch.empty(M, N, dtype=torch.float32, device=vs[0].device) for _ in range(G)]
        px, k1 = _ptr_arr(vs)
        pw, k2 = _ptr_arr(Ws)
        py, k3 = _ptr_arr(ys)
        call('dig3d_smalln_fwd_grouped', G, px, pw, None, M, K, N, py, _stream())
        ctx.G = G
        ctx.pos_only = bool(ops._twice_differentiable)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*vs, *Ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        G = ctx.G
        sv = ctx.saved_tensors
        vs, Ws = sv[:G], sv[G:]
        M, K = vs[0].shape
        N = Ws[0].size(0)
        if all(g is None for g in gys):
            return (None,) * (1 + 2 * G)
        gys = [(_c(g) if g is not None else torch.zeros(M, N, dtype=torch.float32, device=vs[0].device)) for g in gys]
        if torch.is_grad_enabled():
            if not ctx.pos_only:
                raise NotImplementedError('dig_amd heads: a create_graph backward is supported for the position gradient '
                                          'of an energy_and_force forward only')
            gvs = _HeadsBwd2.apply(G, *gys, *Ws)
            return (None,) + tuple(gvs) + (None,) * G
        gxs, gw = _smalln_bwd(gys, Ws, vs, True, Ws)
        N_, K_ = N, K
        return (None,) + tuple(gxs) + tuple((b[:N_ * K_].view(N_, K_) if mine else None) for b, mine in gw)


class _HeadsBwd2(Function):
    """gv_g = gy_g W_g as a differentiable function of (gy_g, W_g)."""

    @staticmethod
    def forward(ctx, G, *tensors):
        gys = [_c(t) for t in tensors[:G]]
        Ws = [_c(t) for t in tensors[G:2 * G]]
        M = gys[0].size(0)
        K = Ws[0].size(1)
        gxs, _ = _smalln_bwd(gys, Ws, [None] * G, True, None)     # input-gradient half only (was: a zero-filled X operand)
        ctx.G = G
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*gys, *Ws)
        return tuple(gxs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *ggvs):
        G = ctx.G
        sv = ctx.saved_tensors
        gys, Ws = sv[:G], sv[G:]
        M, N = gys[0].shape
        K = Ws[0].size(1)
        dev = gys[0].device
        if all(g is None for g in ggvs):
            return (None,) * (1 + 2 * G)
        ggvs = [(_c(g) if g is not None else torch.zeros(M, K, dtype=torch.float32, device=dev)) for g in ggvs]
        ggy = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)]
        px, k1 = _ptr_arr(ggvs)
        pw, k2 = _ptr_arr(Ws)
        py, k3 = _ptr_arr(ggy)
        call('dig3d_smalln_fwd_grouped', G, px, pw, None, M, K, N, py, _stream())
        _, gw = _smalln_bwd(list(gys), Ws, ggvs, False, Ws)
        return (None,) + tuple(ggy) + tuple((b[:N * K].view(N, K) if mine else None) for b, mine in gw)


def heads2_supported(vs, Ws):
    from . import ops
    v0, W0 = vs[0], Ws[0]
    return (ops._twice_differentiable and 1 <= len(vs) <= 8 and v0.is_cuda and v0.dim() == 2 and v0.dtype == torch.float32
            and v0.size(0) > 0 and 1 <= W0.size(0) <= 8 and all(v.shape == v0.shape for v in vs)
            and all(W.shape == W0.shape and W.is_leaf for W in Ws))


def heads2(vs, Ws):
    """[v_g W_g^T] for the G output blocks (out_channels <= 8), twice differentiable, one launch per pass."""
    return list(_Heads2.apply(len(vs), *vs, *Ws))
