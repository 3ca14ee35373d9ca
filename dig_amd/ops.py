"""Operator layer of the engine: thin torch wrappers over the C ABI (include/dig3d.h) plus the
autograd glue.  Mirrors the third-party call signatures the reference uses:

    radius_graph(x, r, batch, loop, max_num_neighbors)      torch_cluster
    scatter(src, index, dim, out, dim_size, reduce)         torch_scatter
    scatter_min(src, index, dim, out, dim_size)             torch_scatter

Everything here launches HIP kernels on the current torch stream; there is no CPU code path.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _hip
from ._hip import call, ptr
from .graph import build_graph, csr_by_key, _stream


def _f32c(t):
    if t.dtype != torch.float32:
        raise RuntimeError(f'expected float32 tensor, got {t.dtype}')
    if not t.is_cuda:
        raise _hip.Dig3dError('dig_amd op received a CPU tensor; the engine has no CPU fallback')
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------
# raw kernels (no autograd)
# ---------------------------------------------------------------------------------------------------
def segment_fused_raw(X, ix, A, B, seg, C):
    """out[s] = sum_{t in seg(s)} A[t] * X[ix[t]] * B[t]  (X/ix, B optional)."""
    ref = A if A is not None else X
    out = torch.empty(seg.S, C, dtype=torch.float32, device=ref.device)
    call('dig3d_segment_fused', ptr(X), ptr(ix), ptr(A), ptr(B), ptr(seg.kptr), ptr(seg.perm), seg.S, C, ptr(out),
         _stream())
    return out


def gather_mul_raw(X, ix, A=None, B=None, cnt=None):
    M, C = ix.numel(), X.size(1)
    out = torch.empty(M, C, dtype=torch.float32, device=X.device)
    call('dig3d_gather_mul', ptr(X), ptr(ix), ptr(A), ptr(B), M, C, ptr(out), ptr(cnt), _stream())
    return out


# ---------------------------------------------------------------------------------------------------
# differentiable primitives — mutually adjoint, closed under differentiation (double backward works,
# which the energy_and_force path needs: run.py:126 differentiates through d out / d pos)
# ---------------------------------------------------------------------------------------------------
class _SegSum(Function):
    @staticmethod
    def forward(ctx, src, seg):
        ctx.seg = seg
        src = _f32c(src)
        return segment_fused_raw(None, None, src, None, seg, src.size(1))

    @staticmethod
    def backward(ctx, g):
        return _Gather.apply(g, ctx.seg), None


class _Gather(Function):
    @staticmethod
    def forward(ctx, x, seg):
        ctx.seg = seg
        return gather_mul_raw(_f32c(x), seg.key, cnt=seg.cnt)

    @staticmethod
    def backward(ctx, g):
        return _SegSum.apply(g, ctx.seg), None


class _SegMean(Function):
    """per-segment mean (scatter(..., reduce='mean')): the CSR segment kernel with the division by the segment length
    fused into its store; backward = row gather of g / count."""

    @staticmethod
    def forward(ctx, src, seg):
        ctx.seg = seg
        src = _f32c(src)
        C = src.size(1)
        out = torch.empty(seg.S, C, dtype=torch.float32, device=src.device)
        call('dig3d_segment_fused_mean', None, None, ptr(src), None, ptr(seg.kptr), ptr(seg.perm), seg.S, C, ptr(out),
             _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        seg = ctx.seg
        g = _f32c(g)
        gs = torch.empty_like(g)
        call('dig3d_rows_div_count', ptr(g), ptr(seg.kptr), seg.S, g.size(1), ptr(gs), _stream())
        return _Gather.apply(gs, seg), None


def segment_mean(src, seg):
    squeeze = src.dim() == 1
    if squeeze:
        src = src.unsqueeze(1)
    out = _SegMean.apply(src, seg)
    return out.squeeze(1) if squeeze else out


def segment_sum(src, seg):
    """[M, C] -> [S, C] sum over the segments of ``seg`` (rows may be non-contiguous if seg.perm)."""
    squeeze = src.dim() == 1
    if squeeze:
        src = src.unsqueeze(1)
    out = _SegSum.apply(src, seg)
    return out.squeeze(1) if squeeze else out


def gather_rows(x, seg):
    """x[seg.key] with a segment-sum backward (no atomics)."""
    squeeze = x.dim() == 1
    if squeeze:
        x = x.unsqueeze(1)
    out = _Gather.apply(x, seg)
    return out.squeeze(1) if squeeze else out


class _EdgeCat(Function):
    """cat([x[i], x[j], r], -1) (spherenet.py:88-89) in one launch; backward in one more (csrc/segment.hip:dig3d_edge_cat)."""

    @staticmethod
    def forward(ctx, x, r, seg_i, seg_j):
        x, r = _f32c(x), _f32c(r)
        E, Cx, Cr = r.size(0), x.size(1), r.size(1)
        out = torch.empty(E, 2 * Cx + Cr, dtype=torch.float32, device=x.device)
        call('dig3d_edge_cat', ptr(x), ptr(seg_i.key), ptr(seg_j.key), ptr(r), E, Cx, Cr, ptr(out), ptr(seg_i.cnt), _stream())
        ctx.seg_i, ctx.seg_j, ctx.dims = seg_i, seg_j, (x.size(0), E, Cx, Cr)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        N, E, Cx, Cr = ctx.dims
        si, sj = ctx.seg_i, ctx.seg_j
        G = _f32c(G)
        gx = torch.empty(N, Cx, dtype=torch.float32, device=G.device)
        gr = torch.empty(E, Cr, dtype=torch.float32, device=G.device)
        call('dig3d_edge_cat_bwd', ptr(G), ptr(si.kptr), ptr(si.perm), ptr(sj.kptr), ptr(sj.perm), N, E, Cx, Cr, ptr(gx),
             ptr(gr), _stream())
        return gx, gr, None, None


class _EdgeCatEmb(Function):
    """cat([emb[z][i], emb[z][j], r], -1): the embedding lookup folded into the edge_cat launch (no [N, C] node tensor, one
    launch less per step); backward = dig3d_edge_cat_bwd then the embedding's weight gradient (``_Embedding.backward``)."""

    @staticmethod
    def forward(ctx, z, weight, r, seg_i, seg_j):
        weight, r = _f32c(weight), _f32c(r)
        E, Cx, Cr = r.size(0), weight.size(1), r.size(1)
        out = torch.empty(E, 2 * Cx + Cr, dtype=torch.float32, device=r.device)
        call('dig3d_edge_cat_emb', ptr(weight), ptr(z), ptr(seg_i.key), ptr(seg_j.key), ptr(r), E, Cx, Cr, ptr(out),
             ptr(seg_i.cnt), _stream())
        ctx.seg_i, ctx.seg_j, ctx.dims = seg_i, seg_j, (z.numel(), E, Cx, Cr)
        ctx.save_for_backward(z)
        ctx.shape = weight.shape
        ctx.leaf = weight.is_leaf and not _twice_differentiable
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        gx, gr, _, _ = _EdgeCat.backward(ctx, G)
        _, gW = _Embedding.backward(ctx, gx)
        return None, gW, gr, None, None


def edge_cat_emb_supported(z, weight, r, seg_i, seg_j):
    return (_embed_kernel and z.is_cuda and z.dim() == 1 and z.dtype == torch.int64 and z.is_contiguous()
            and weight.dtype == torch.float32 and weight.dim() == 2 and weight.size(0) <= 128 and not _twice_differentiable
            and r.is_cuda and r.dim() == 2 and r.dtype == torch.float32 and seg_i.S == z.numel() == seg_j.S
            and bool(_hip.query('dig3d_edge_cat_supported', weight.size(1), r.size(1))))


def edge_cat_emb(z, weight, r, seg_i, seg_j):
    """``torch.cat([emb(z)[i], emb(z)[j], r], -1)`` in one launch (see ``_EdgeCatEmb``)."""
    return _EdgeCatEmb.apply(z, weight, r, seg_i, seg_j)


def edge_cat(x, r, seg_i, seg_j):
    """``torch.cat([x[i], x[j], r], dim=-1)`` with i = seg_i.key, j = seg_j.key (the edge initialisation's input)."""
    if (x.is_cuda and not _twice_differentiable and x.dim() == 2 and r.dim() == 2 and x.dtype == r.dtype == torch.float32
            and seg_i.S == x.size(0) == seg_j.S and _hip.query('dig3d_edge_cat_supported', x.size(1), r.size(1))):
        return _EdgeCat.apply(x, r, seg_i, seg_j)
    return torch.cat([gather_rows(x, seg_i), gather_rows(x, seg_j), r], dim=-1)


class _GatherMulSegSum(Function):
    """out[s] = sum_{t in seg_out(s)} X[gat.key[t]] * A[t] * B[t]   (B optional) — first-order fused op.
    Forward + three backward kernels instead of gather, 2 multiplies, scatter_add and their adjoints."""

    @staticmethod
    def forward(ctx, X, A, B, gat, seg_out):
        X, A = _f32c(X), _f32c(A)
        B = _f32c(B) if B is not None else None
        ctx.gat, ctx.seg_out = gat, seg_out
        ctx.save_for_backward(X, A, B)
        return segment_fused_raw(X, gat.key, A, B, seg_out, X.size(1))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        X, A, B = ctx.saved_tensors
        gat, seg_out = ctx.gat, ctx.seg_out
        G = _f32c(G)
        C = X.size(1)
        gX = gA = gB = None
        if ctx.needs_input_grad[0]:
            # gX[e'] = sum_{t: gat.key[t] = e'} G[seg_out.key[t]] * A[t] * B[t]
            gX = segment_fused_raw(G, seg_out.key, A, B, gat, C)
        if ctx.needs_input_grad[1] or (B is not None and ctx.needs_input_grad[2]):
            M = A.size(0)
            gA = torch.empty_like(A) if ctx.needs_input_grad[1] else None
            gB = torch.empty_like(A) if (B is not None and ctx.needs_input_grad[2]) else None
            # rows >= cnt (padding of a static-shape batch) get exact zeros: their factor rows feed weight gradients
            call('dig3d_gather_mul2', ptr(G), ptr(seg_out.key), ptr(X), ptr(gat.key), ptr(A), ptr(B), M, C,
                 ptr(gA), ptr(gB), ptr(seg_out.cnt if seg_out.cnt is not None else gat.cnt), _stream())
        return gX, gA, gB, None, None


def gather_mul_segment_sum(X, A, B, gat, seg_out, composite=False):
    """sum_{t in seg_out(s)} X[gat.key[t]] * A[t] * B[t].  ``composite=True`` builds it from the
    gather / segment-sum primitives (differentiable to any order — the energy_and_force path);
    otherwise one fused first-order kernel."""
    if composite and X.size(1) % 4 == 0:
        # closed family of two kernels (dig_amd/diffops.py): differentiable to any order without elementwise glue
        from . import diffops
        return diffops.gather_mul_segsum(X, A if B is None else A * B, gat, seg_out)
    if composite or X.size(1) % 4 != 0:
        # generic-width composition (differentiable to any order)
        m = gather_rows(X, gat) * A
        if B is not None:
            m = m * B
        return segment_sum(m, seg_out)
    return _GatherMulSegSum.apply(X, A, B, gat, seg_out)


class _Embedding(Function):
    """weight[idx] (nn.Embedding forward) with the gradient w.r.t. ``weight`` from csrc/segment.hip:k_embedding_bwd_part
    (LDS tables per 256-row chunk, then one reduction): deterministic, 3-10x faster than the framework's sort-based
    kernel at 600 - 16 000 atoms."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.shape = weight.shape
        ctx.leaf = weight.is_leaf and not _twice_differentiable
        weight = _f32c(weight)
        out = torch.empty(idx.numel(), weight.size(1), dtype=torch.float32, device=weight.device)
        call('dig3d_embedding_fwd', ptr(idx), ptr(weight), idx.numel(), weight.size(0), weight.size(1), ptr(out), _stream())
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        V, C = ctx.shape
        g = _f32c(g)
        M = idx.numel()
        nch = _hip.query('dig3d_embedding_bwd_chunks', M)
        part = torch.empty(nch * V * C, dtype=torch.float32, device=g.device)
        gW = torch.empty(V, C, dtype=torch.float32, device=g.device)
        # the sum of the chunk tables joins the step's one reduction launch when there is one (deferred_reductions)
        now = 1 if M == 0 else _reduce_later(part, nch, V * C, gW, ctx.leaf)
        call('dig3d_embedding_bwd', ptr(idx), ptr(g), M, V, C, ptr(part), ptr(gW), now, _stream())
        return None, gW


def embedding(idx, weight):
    """``nn.Embedding`` lookup for a 1-D int64 index on the GPU (atom types, <= 128 of them); the framework op otherwise."""
    if (_embed_kernel and idx.is_cuda and idx.dim() == 1 and idx.dtype == torch.int64 and weight.dtype == torch.float32
            and idx.is_contiguous() and weight.size(0) <= 128):
        C = weight.size(1)
        if C % 4:                  # float4 rows: zero-padded columns in, sliced back out (hidden_channels = 50, ...)
            return pad2d(_Embedding.apply(idx, pad2d(weight, weight.size(0), (C + 3) & ~3)), idx.numel(), C)
        return _Embedding.apply(idx, weight)
    return torch.nn.functional.embedding(idx, weight)


class _FeatConv(Function):
    """out[s] = sum_{t in seg_out(s)} X[gat.key[t]] * (Wc f_t): ComENet's EdgeGraphConv aggregation with the edge weight
    (a linear map of <= 16 edge features) evaluated inside the kernel — csrc/segment.hip:k_featconv; forward, the
    gradient w.r.t. X (same kernel, transposed CSR) and the gradient w.r.t. Wc (k_featconv_wgrad) never form the
    [E, C] edge-weight tensor."""

    @staticmethod
    def forward(ctx, X, F, Wc, gat, seg_out, tap=False):
        X, F, Wc = _f32c(X), _f32c(F), _f32c(Wc)
        C, K = X.size(1), F.size(1)
        out = torch.empty(seg_out.S, C, dtype=torch.float32, device=X.device)
        call('dig3d_featconv', ptr(X), ptr(gat.key), ptr(F), K, ptr(Wc), ptr(seg_out.kptr), ptr(seg_out.perm), seg_out.S,
             C, ptr(out), None, _stream())
        ctx.gat, ctx.seg_out, ctx.tap = gat, seg_out, tap
        ctx.save_for_backward(X, F, Wc)
        # tap: also hand back an alias of X for its OTHER consumers — their gradient arrives here as an argument and is
        # added by the kernel that produces this op's own input gradient (no framework addition)
        return (out, X.view_as(X)) if tap else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G, ga=None):
        X, F, Wc = ctx.saved_tensors
        gat, seg_out = ctx.gat, ctx.seg_out
        G = _f32c(G)
        C, K = X.size(1), F.size(1)
        gX = gW = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty(gat.S, C, dtype=torch.float32, device=X.device)
            rows, feats, perm = seg_out.key, F, gat.perm
            if perm is not None and perm.numel() == F.size(0) and gat.cnt is None:   # (a static-shape graph's buffers are
                # refilled in place every step: nothing derived from them may be cached on the segmentation)
                # the transposed direction walks the edges through `perm`: per edge a dependent perm -> row-index load
                # and a feature row from a scattered address (131 us per launch against 66 in edge order at 5e5 edges).
                # The graph and the features are the same for every layer of the step: permute both ONCE into this
                # grouping's order and run the edge-order form.
                if gat.aux is None:
                    gat.aux = {}
                k = ('featconv', F.data_ptr(), seg_out.key.data_ptr())
                if k not in gat.aux:
                    gat.aux[k] = (gather_mul_raw(seg_out.key.view(torch.float32).unsqueeze(1), perm).view(torch.int32).squeeze(1),
                                  gather_mul_raw(F, perm), F)       # F kept: the key is its address
                rows, feats, perm = gat.aux[k][0], gat.aux[k][1], None
            call('dig3d_featconv', ptr(G), ptr(rows), ptr(feats), K, ptr(Wc), ptr(gat.kptr), ptr(perm), gat.S, C,
                 ptr(gX), ptr(_f32c(ga)) if ga is not None else None, _stream())
        if ctx.needs_input_grad[2]:
            M = F.size(0)
            nb = _hip.query('dig3d_featconv_wgrad_blocks', M)
            part = torch.empty(nb * C * K, dtype=torch.float32, device=X.device)
            gW = torch.empty(C, K, dtype=torch.float32, device=X.device)
            call('dig3d_featconv_wgrad', ptr(G), ptr(seg_out.key), ptr(X), ptr(gat.key), ptr(F), K, M, C, ptr(part),
                 ptr(gW), 1, ptr(seg_out.cnt), _stream())       # (cnt: the live edges of a static-shape batch)
        return gX, None, gW, None, None, None


def feature_conv_supported(X, F, Wc):
    return (X.is_cuda and X.dim() == 2 and X.dtype == torch.float32 and F.dim() == 2 and not F.requires_grad
            and not _twice_differentiable and Wc.shape == (X.size(1), F.size(1))
            and bool(_hip.query('dig3d_featconv_supported', F.size(1), X.size(1))))


def feature_conv(X, F, Wc, gat, seg_out, tap=False):
    """sum_{t in seg_out(s)} X[gat.key[t]] * (F[t] Wc^T)  (comenet.py:130-133 with edge_weight = lin_feature(feature)).
    ``tap=True``: -> (result, alias of X): use the alias for every other consumer of X and their gradients are added inside
    this op's backward kernel instead of by autograd."""
    return _FeatConv.apply(X, F, Wc, gat, seg_out, tap)


# ---------------------------------------------------------------------------------------------------
# dense hidden-channel layers on the f32 matrix cores (csrc/dense.hip)
# ---------------------------------------------------------------------------------------------------
ACT_NONE, ACT_SWISH, ACT_SSP = 0, 1, 2
_ACT_DERIV, _ACT_KEEP_DERIV = 3, 4     # csrc/dense_common.h: the saved tensor of a once-differentiated layer is act'(z)
_twice_differentiable = False      # set by the models on the energy_and_force path (double backward)


class composite_mode:
    """``with composite_mode(True):`` — every op takes its twice-differentiable form (dig_amd/diffops.py Functions +
    HIP gather/segment primitives).  Needed only when forces are trained (run.py:126 double backward)."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global _twice_differentiable
        self.prev, _twice_differentiable = _twice_differentiable, self.on

    def __exit__(self, *a):
        global _twice_differentiable
        _twice_differentiable = self.prev


_deferred = None      # the active ``deferred_reductions`` block (dig_amd/graphed.py), or None


class deferred_reductions:
    """``with deferred_reductions() as d: grads = autograd.grad(...)`` then ``d.flush()``: inside, every dense backward
    writes only its per-worker partial gradients; ``flush`` reduces ALL layers in one launch (79 reductions of ~5 us
    each per SphereNet step otherwise).  The returned weight gradients are valid only after ``flush``.

    A weight that enters the autograd graph more than once (the force path: forward node + double-backward node of the
    same layer) registers every contribution under the weight's key: the first registrant owns the gradient buffer,
    later ones return ``None`` to autograd and their partials are ADDED by a second launch — no per-layer reduction
    and no autograd add kernel per weight."""

    def __enter__(self):
        global _deferred
        self.prev, _deferred = _deferred, self
        self.items, self.by_key, self.wgrads, self.after = [], {}, [], []
        return self

    def __exit__(self, *a):
        global _deferred
        _deferred = self.prev

    def add(self, part, nb, stride, gwb, n=None):
        self.items.append(dict(out=gwb, stride=stride, parts=[(part, nb, stride if n is None else n)]))

    def add_keyed(self, key, part, nb, stride, n, device, row_stride=None):
        """-> gradient buffer [stride] if this is the first contribution for ``key``, else None.  ``row_stride``: the distance
        between the partial rows of ``part`` when it is a slice of a wider partial buffer (default: ``stride``)."""
        it = self.by_key.get(key)
        if it is None:
            gwb = torch.empty(stride, dtype=torch.float32, device=device)
            it = dict(out=gwb, stride=stride, parts=[(part, nb, n, row_stride)])
            self.by_key[key] = it
            self.items.append(it)
            return gwb
        it['parts'].append((part, nb, n, row_stride))
        return None

    def owner(self, key):
        it = self.by_key.get(key)
        return None if it is None else it['out']

    def add_wgrad(self, GY, X, K, N, Z=None, act=0, key=None, n_valid=None):
        """defer a whole weight-gradient GEMM gW = (GY * act'(Z))^T X (+ bias column sums): the dense layers of the backward
        pass become ONE launch at ``flush`` (csrc/dense.hip:dig3d_wgrad_many; a few launches beyond 64 tiles of 128 x 128)
        with a worker count chosen for the whole set — 44 layers x 11 workers for the default SphereNet instead of 8
        launches writing 32 - 85 partials per layer.  -> the gradient buffer float[N*K + N] (valid after ``flush``).

        ``key`` (the weight's data pointer; energy_and_force, where a weight enters the second-order graph more than once):
        -> (buffer, mine) — the first contribution under a key owns the buffer and hands it to autograd (``mine``), later
        ones are ADDED to it at flush, their first ``n_valid`` floats (N*K: a contribution without a bias part)."""
        stride = N * K + N
        zz, aa = (Z if act != 0 else None), (act if Z is not None else 0)
        if key is None:
            gwb = torch.empty(stride, dtype=torch.float32, device=GY.device)
            self.wgrads.append((GY, X, K, N, gwb, zz, aa, None, stride))
            return gwb
        it = self.by_key.get(key)
        mine = it is None
        if mine:
            it = dict(out=torch.empty(stride, dtype=torch.float32, device=GY.device), stride=stride, parts=[])
            self.by_key[key] = it
            self.items.append(it)
        self.wgrads.append((GY, X, K, N, it['out'], zz, aa, it, stride if n_valid is None else n_valid))
        return it['out'], mine

    def _flush_wgrads(self):
        ws, self.wgrads = self.wgrads, []
        dev = ws[0][0].device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        tiles = lambda w: -(-w[3] // 128) * -(-w[2] // 128)
        # launches of <= 64 tiles (the kernel's table); inside one, every layer gets workers in proportion to its rows
        # (r06, measured and not kept: a per-layer table with one word per tile — 128 tiles, a whole SphereNet pass in ONE
        # launch instead of 64 + 50 tiles: config 2 1.455 -> 1.516 ms, config 4 5.15 -> 5.41, config 5 6.84 -> 6.88 on one box;
        # with all tiles sharing the two-blocks-per-CU budget every edge layer gets fewer, longer workers and the launch
        # lasts as long as its longest block.  Also found there: unsigned char / short arrays in a by-value kernel argument,
        # indexed with blockIdx, faulted on gfx950 / ROCm 7.2 — whole words did not.)
        # so that the launch is ~2 blocks per CU of equal work (a uniform count starves the long layers when the output
        # blocks' 600-row layers share a launch with the 8 700-row edge layers)
        chunks, cur, nt = [], [], 0
        for w in ws:
            if cur and nt + tiles(w) > 64:
                chunks.append(cur)
                cur, nt = [], 0
            cur.append(w)
            nt += tiles(w)
        chunks.append(cur)
        for chunk in chunks:
            n = len(chunk)
            cost = sum(tiles(w) * w[0].size(0) for w in chunk)
            # rows per worker: at least 256 (a worker writes a 64-KB partial tile per 128 x 128 tile), and large enough that
            # the launch fits the chip in ONE pass of two blocks per CU (a few blocks over cost a whole second pass)
            cap = max(1, int(wgrad_blocks_per_cu * cus))
            rpw = max(256, -(-cost // cap))
            while True:
                nws = [max(1, min(64, -(-w[0].size(0) // rpw))) for w in chunk]
                if sum(k * tiles(w) for w, k in zip(chunk, nws)) <= cap or rpw >= 1 << 20:
                    break
                rpw += max(1, rpw // 16)
            parts = [torch.empty(k * (w[3] * w[2] + w[3]), dtype=torch.float32, device=dev) for w, k in zip(chunk, nws)]
            PP, IA = ctypes.c_void_p * n, ctypes.c_int * n
            cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
            call('dig3d_wgrad_many', n, cast(PP(*[ptr(w[0]) for w in chunk])), cast(PP(*[ptr(w[5]) for w in chunk])),
                 cast(IA(*[w[6] for w in chunk])), cast(PP(*[ptr(w[1]) for w in chunk])), cast(IA(*[w[2] for w in chunk])),
                 cast(IA(*[w[3] for w in chunk])), cast(IA(*[w[0].size(0) for w in chunk])), cast(IA(*nws)),
                 cast(PP(*[ptr(t) for t in parts])), int(wgrad_double_buffer), _stream())
            for w, part, k in zip(chunk, parts, nws):
                if w[7] is None:
                    self.add(part, k, w[3] * w[2] + w[3], w[4])
                else:                                   # a keyed contribution: joins its weight's rounds
                    w[7]['parts'].append((part, k, w[8]))

    @staticmethod
    def _launch(name, rows):
        n = len(rows)
        if n == 0:
            return
        PP, IA, LA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_int64 * n
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        call(name, cast(PP(*[ptr(r[0]) for r in rows])), cast(IA(*[r[1] for r in rows])), cast(LA(*[r[2] for r in rows])),
             cast(IA(*[r[3] for r in rows])), cast(PP(*[ptr(r[4]) for r in rows])), n, _stream())

    def flush(self):
        if self.wgrads:
            self._flush_wgrads()
        # round 0 WRITES every gradient from its widest contribution (weights + bias); round k >= 1 ADDS the k-th further
        # contribution of every gradient that has one.  One launch per round: two contributions to the same gradient inside
        # one accumulating launch would be a read-modify-write race between their blocks (a weight of the closed triplet
        # family enters the second-order graph three times: diffops.trip2)
        rounds = []
        for it in self.items:
            parts = sorted(it['parts'], key=lambda p: -p[2])
            for k, pp in enumerate(parts):
                part, nb, n = pp[0], pp[1], pp[2]
                rs = pp[3] if len(pp) > 3 and pp[3] is not None else it['stride']
                while len(rounds) <= k:
                    rounds.append([])
                rounds[k].append((part, nb, rs, n, it['out']))
        for k, rows in enumerate(rounds):
            self._launch('dig3d_reduce_many' if k == 0 else 'dig3d_reduce_many_acc', rows)
        self.items, self.by_key = [], {}
        for fn in self.after:              # gradients assembled from reduced pieces (linear_cat2)
            fn()
        self.after = []


def backward(loss, params):
    """``loss.backward()`` of an eager step with the weight-gradient reductions of ALL layers deferred into one launch
    (``deferred_reductions``): gradients are ASSIGNED to ``p.grad`` (added when one is already there).  ComENet at
    config 5: 51 per-layer reductions (0.6 ms of an 10.8 ms step) become one."""
    params = [p for p in params if p.requires_grad]
    with deferred_reductions() as red:
        grads = torch.autograd.grad(loss, params, allow_unused=True)
    red.flush()
    for p, g in zip(params, grads):
        if g is None:
            continue
        p.grad = g if p.grad is None else p.grad + g


def _all_leaf(ts):
    """True when every weight is an autograd leaf.  A COMPUTED weight (ComENet's ``lin2.weight @ lin1.weight``,
    comenet.py:105) is differentiated further during the very backward pass that produces its gradient, so that gradient
    must be complete when the Function returns: its reduction cannot wait for ``deferred_reductions.flush``."""
    return all(t is None or t.is_leaf for t in ts)


def _reduce_later(part, nb, stride, gwb, leaf=True):
    """-> reduce_now flag for the C call; registers the reduction when a deferred_reductions block is active (and the
    weight is a leaf: see ``_all_leaf``)."""
    if _deferred is None or not leaf:
        return 1
    _deferred.add(part, nb, stride, gwb)
    return 0


class _LinearAct(Function):
    """y = act(x W^T + b) (+ res): one MFMA kernel forward; backward = dgrad + wgrad kernels with act' applied on
    the fly (the pre-activation is the only extra tensor kept)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act, rowscale=None):
        x, weight = _f32c(x), _f32c(weight)
        M, K = x.shape
        N = weight.size(0)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        if rowscale is not None:
            # y = rowscale[m] * (x W^T + b): the multiplier, broadcast over the columns, is this layer's "derivative tensor"
            z = torch.empty_like(y)
            call('dig3d_linear_fwd_rowscale', ptr(x), ptr(weight), ptr(bias), ptr(_f32c(rowscale)), M, K, N, ptr(y), ptr(z),
                 _stream())
            ctx.small, ctx.leaf, ctx.res_is_x = False, _all_leaf((weight, bias)), False
            ctx.save_for_backward(x, weight, z)
            ctx.act, ctx.has_bias, ctx.has_res = _ACT_DERIV, bias is not None, False
            return y
        z = torch.empty_like(y) if act != ACT_NONE else None
        small = K <= 16 and N <= 256       # radial-basis / feature projections: dedicated no-tile kernels (csrc/dense.hip)
        # this Function is differentiated once: the forward keeps act'(z) (same bytes as z), the input- and weight-gradient
        # kernels of the backward multiply by it instead of each re-evaluating the exponential
        call('dig3d_smallk_fwd' if small else 'dig3d_linear_fwd', ptr(x), ptr(weight), ptr(bias),
             ptr(res.contiguous() if res is not None else None), M, K, N,
             act | _ACT_KEEP_DERIV if act != ACT_NONE else act, ptr(y), ptr(z), _stream())
        ctx.small = small
        ctx.leaf = _all_leaf((weight, bias))
        ctx.save_for_backward(x, weight, z)
        ctx.act, ctx.has_bias, ctx.has_res = act, bias is not None, res is not None
        # y = x + act(lin(x)) (the residual layers of ComENet, comenet.py:208-209): the input gradient kernel adds gy itself
        # instead of autograd adding the two gradients of x afterwards (one 3-tensor elementwise launch per layer)
        ctx.res_is_x = (res is not None and K == N and res.data_ptr() == x.data_ptr() and res.shape == x.shape
                        and res.is_contiguous() and res.dtype == torch.float32)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight, z = ctx.saved_tensors
        gy = _f32c(gy)
        M, K = x.shape
        N = weight.size(0)
        st = _stream()
        gx = gw = gb = None
        want_x = ctx.needs_input_grad[0]
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if want_x:
            gx = torch.empty_like(x)
        bact = _ACT_DERIV if ctx.act != ACT_NONE else ACT_NONE      # z holds act'(pre-activation)
        fold = ctx.res_is_x and want_x                  # gx = gy + (gy * act'(z)) W in the kernel; nothing for `res`
        gadd = ptr(gy) if fold else None
        tap = getattr(ctx, 'tap_add', None)             # _LinearTap: the gradient that reached the alias of x
        if tap is not None:
            gadd = ptr(tap)
        # big layers only (ComENet's 16 384 x 256 x 256: 78 us merged vs ~30 + ~25 split; config 5 10.2 -> 9.2 ms): at
        # SphereNet's 8.7k x 128 x 384 edge-initialisation layer the merged launch is the cheaper one (47 vs 25 + 47 us)
        defer = (want_w and not ctx.small and _deferred is not None and ctx.leaf and M > 0 and (K & 3) == 0
                 and (N & 3) == 0 and M * N * K >= linear_defer_min)
        if want_w and not defer:
            nb = _hip.query('dig3d_smallk_blocks' if ctx.small else 'dig3d_linear_wgrad_blocks', M)
            part = torch.empty(nb * (N * K + N), dtype=torch.float32, device=x.device)
            gwb = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
            gw = gwb[:N * K].view(N, K)
            gb = gwb[N * K:] if ctx.has_bias else None
        stride = N * K + N
        if ctx.small:
            if want_x or want_w:
                now = _reduce_later(part, nb, stride, gwb, ctx.leaf) if want_w else 1
                call('dig3d_smallk_bwd', ptr(gy), ptr(z), ptr(weight), ptr(x), M, K, N, bact, ptr(gx), gadd,
                     ptr(part) if want_w else None, ptr(gwb) if want_w else None, now, st)
        elif defer:
            # inside a deferred_reductions block: the input gradient now, the weight gradient in the one launch that
            # covers every dense layer of the backward pass (dig3d_wgrad_many)
            if want_x:
                call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(weight), M, K, N, bact, ptr(gx), gadd, st)
            gwb = _deferred.add_wgrad(gy, x, K, N, z, bact)
            gw = gwb[:N * K].view(N, K)
            gb = gwb[N * K:] if ctx.has_bias else None
        elif want_x and want_w:      # one launch: weight-gradient workers + input-gradient row tiles
            now = _reduce_later(part, _hip.query('dig3d_linear_bwd_workers', M, K, N), stride, gwb, ctx.leaf)
            call('dig3d_linear_bwd', ptr(gy), ptr(z), ptr(weight), ptr(x), M, K, N, bact, ptr(gx), gadd, ptr(part),
                 ptr(gwb), now, st)
        elif want_x:
            call('dig3d_linear_bwd_input', ptr(gy), ptr(z), ptr(weight), M, K, N, bact, ptr(gx), gadd, st)
        elif want_w:
            now = _reduce_later(part, nb, stride, gwb, ctx.leaf)
            call('dig3d_linear_bwd_weight', ptr(gy), ptr(z), ptr(x), M, K, N, bact, ptr(part), ptr(gwb), now, st)
        return gx, gw, gb, (gy if ctx.has_res and not fold else None), None, None


class _LinearTap(Function):
    """(act(x W^T + b), alias of x): the alias is for the OTHER consumers of x (a skip connection, a second layer) — their
    gradient comes back as an argument of this backward and is added by the input-gradient kernel (`gx_add`), so autograd
    never launches an addition for x (schnet.py:34 + :56-59: v feeds `lin` and the residual; comenet.py:146-147)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _f32c(x)
        y = _LinearAct.forward(ctx, x, weight, bias, None, act)
        return y, x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy, ga):
        ctx.tap_add = _f32c(ga) if ga is not None else None
        if gy is None:
            gy = torch.zeros(ctx.saved_tensors[0].size(0), ctx.saved_tensors[1].size(0), dtype=torch.float32,
                             device=ctx.saved_tensors[0].device)
        gx, gw, gb = _LinearAct.backward(ctx, gy)[:3]
        if gx is None and ga is not None and ctx.needs_input_grad[0]:
            gx = ga
        return gx, gw, gb, None


def linear_tap(x, weight, bias=None, act=ACT_NONE):
    """``(linear(x, weight, bias, act), x')`` with x' an alias of x whose gradient is folded into this layer's input-gradient
    kernel; plain ``(linear(...), x)`` where the MFMA layer does not apply."""
    N = weight.size(0)
    if (x.is_cuda and not _twice_differentiable and x.dim() == 2 and x.dtype == weight.dtype == torch.float32 and (N & 7) == 0
            and x.size(0) > 0 and x.requires_grad and act in (ACT_NONE, ACT_SWISH, ACT_SSP)):
        return _LinearTap.apply(x, weight, bias, act)
    return linear(x, weight, bias, act), x


class _LinearCat2(Function):
    """y = cat([h1, h2], 1) W^T + b (+ res) without the concatenation (comenet.py:199-200): two launches of the persistent
    kernel on the column halves of W, the second adding to the first; backward = two input-gradient launches (two
    contiguous tensors, no slicing copies) and the two weight-gradient halves."""

    @staticmethod
    def forward(ctx, h1, h2, weight, bias, res):
        h1, h2, weight = _f32c(h1), _f32c(h2), _f32c(weight)
        M, H = h1.shape
        N = weight.size(0)
        t = torch.empty(M, N, dtype=torch.float32, device=h1.device)
        y = torch.empty_like(t)
        st = _stream()
        call('dig3d_linear_fwd_wslice', ptr(h1), ptr(weight), 2 * H, None, ptr(res.contiguous() if res is not None else None),
             M, H, N, ACT_NONE, ptr(t), None, st)
        call('dig3d_linear_fwd_wslice', ptr(h2), weight.data_ptr() + 4 * H, 2 * H, ptr(bias), ptr(t), M, H, N, ACT_NONE,
             ptr(y), None, st)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.leaf = _all_leaf((weight, bias))
        ctx.save_for_backward(h1, h2, weight)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        h1, h2, weight = ctx.saved_tensors
        gy = _f32c(gy)
        M, H = h1.shape
        N = weight.size(0)
        st = _stream()
        g1 = g2 = gw = gb = None
        if ctx.needs_input_grad[0]:
            g1 = torch.empty_like(h1)
            call('dig3d_linear_bwd_input_wslice', ptr(gy), None, ptr(weight), 2 * H, M, H, N, ACT_NONE, ptr(g1), None, st)
        if ctx.needs_input_grad[1]:
            g2 = torch.empty_like(h2)
            call('dig3d_linear_bwd_input_wslice', ptr(gy), None, weight.data_ptr() + 4 * H, 2 * H, M, H, N, ACT_NONE,
                 ptr(g2), None, st)
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            gw = torch.empty(N, 2 * H, dtype=torch.float32, device=gy.device)
            halves = []
            for x in (h1, h2):
                if _deferred is not None and ctx.leaf:
                    halves.append(_deferred.add_wgrad(gy, x, H, N))
                else:
                    nb = _hip.query('dig3d_linear_wgrad_blocks', M)
                    part = torch.empty(nb * (N * H + N), dtype=torch.float32, device=gy.device)
                    gwb = torch.empty(N * H + N, dtype=torch.float32, device=gy.device)
                    call('dig3d_linear_bwd_weight', ptr(gy), None, ptr(x), M, H, N, ACT_NONE, ptr(part), ptr(gwb), 1, st)
                    halves.append(gwb)

            def assemble(gw=gw, halves=halves, H=H, N=N):
                gw[:, :H].copy_(halves[0][:N * H].view(N, H))
                gw[:, H:].copy_(halves[1][:N * H].view(N, H))
            gb = halves[0][N * H:] if ctx.has_bias else None
            if _deferred is not None and ctx.leaf:
                _deferred.after.append(assemble)      # the halves are complete after ``flush``
            else:
                assemble()
        return g1, g2, gw, gb, (gy if ctx.has_res else None)


def linear_cat2(h1, h2, weight, bias=None, res=None):
    """``F.linear(torch.cat([h1, h2], 1), weight, bias) (+ res)`` — without the concatenation where the persistent dense
    kernel takes the shape (``dig3d_linear_wslice_supported``), the plain composition otherwise."""
    H = h1.size(1)
    if (h1.is_cuda and not _twice_differentiable and h1.dim() == 2 and h1.shape == h2.shape
            and h1.dtype == h2.dtype == weight.dtype == torch.float32 and weight.size(1) == 2 * H
            and weight.is_contiguous() and (res is None or res.shape == (h1.size(0), weight.size(0)))
            and _hip.query('dig3d_linear_wslice_supported', h1.size(0), H, weight.size(0))):
        return _LinearCat2.apply(h1, h2, weight, bias, res)
    return linear(torch.cat([h1, h2], 1), weight, bias, ACT_NONE, res)


class _Chain(Function):
    """Y_l = res_l + act_l(Y_{l-1} W_l^T + b_l), l < nl <= 8, as ONE forward launch (csrc/chain.hip:k_chainr_fwd: the row
    tile stays on the CU between layers, the weight slices in registers).  Backward: two launches — the input-gradient
    recursion of all layers on the same kind of tile (k_chainr_bwd) and the weight gradients of all layers
    (dense.hip:k_chain_wgrad); the per-layer merged dgrad+wgrad sweep is kept as the reference route
    (``ops._chain_bwd_fused = False``) the fused one is tested against."""

    @staticmethod
    def forward(ctx, x0, spec, packed, *tensors):
        nl = len(spec)
        x0 = _f32c(x0)
        M = x0.size(0)
        dev = x0.device
        Ws = [_f32c(tensors[3 * l]) for l in range(nl)]
        bs = [tensors[3 * l + 1] for l in range(nl)]
        rs = [(_f32c(tensors[3 * l + 2]) if tensors[3 * l + 2] is not None else None) for l in range(nl)]
        Zs = [torch.empty(M, 128, dtype=torch.float32, device=dev) if spec[l][1] != ACT_NONE else None
              for l in range(nl)]
        Ys = [torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(nl)]
        PP, IA = ctypes.c_void_p * nl, ctypes.c_int * nl
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        Ks = cast(IA(*[sp[0] for sp in spec]))
        # the weights re-laid in MFMA operand order, forward and backward formats (csrc/chain.hip:k_chain_pack): every
        # workgroup streams all of them for ~34 rows at E ~ 8.7k, so that stream has to be contiguous kilobyte loads
        if packed is None:          # (a model forward packs all its chains and fronts in one launch: pack_weights)
            packed = pack_weights(Ws)
        call('dig3d_chainp_fwd', ptr(x0), M, nl, ptr(packed[0]), cast(PP(*[ptr(b) for b in bs])),
             cast(PP(*[ptr(r) for r in rs])), cast(PP(*[ptr(z) for z in Zs])), cast(PP(*[ptr(y) for y in Ys])),
             Ks, cast(IA(*[sp[2] for sp in spec])), cast(IA(*[sp[3] for sp in spec])),
             cast(IA(*[sp[1] for sp in spec])), _stream())
        ctx.spec = spec
        ctx.leaf = _all_leaf(Ws) and _all_leaf(bs)
        ctx.has = [(bs[l] is not None, rs[l] is not None) for l in range(nl)]
        ctx.save_for_backward(x0, *Ws, *[z if z is not None else x0.new_empty(0) for z in Zs], *Ys[:-1], packed)
        return Ys[-1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        spec = ctx.spec
        nl = len(spec)
        sv = ctx.saved_tensors
        x0, Ws, Zs, Ys, packed = sv[0], sv[1:1 + nl], sv[1 + nl:1 + 2 * nl], sv[1 + 2 * nl:-1], sv[-1]
        M = x0.size(0)
        dev = x0.device
        st = _stream()
        N = 128
        PP, IA = ctypes.c_void_p * nl, ctypes.c_int * nl
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        Ks = [sp[0] for sp in spec]
        if _chain_bwd_fused:
            # (1) input-gradient recursion of the whole chain on LDS-resident row tiles, (2) all weight gradients in one
            # launch (csrc/dense.hip:k_chain_bwd / k_chain_wgrad)
            GZ = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(nl)]
            gres = [torch.empty(M, N, dtype=torch.float32, device=dev) if spec[l][2] == 1 else None for l in range(nl)]
            gx0 = torch.empty(M, Ks[0], dtype=torch.float32, device=dev)
            zs = [Zs[l] if spec[l][1] != ACT_NONE else None for l in range(nl)]
            call('dig3d_chainp_bwd', ptr(_f32c(gout)), M, nl, ptr(packed[1]), cast(PP(*[ptr(z) for z in zs])),
                 cast(PP(*[ptr(g) for g in GZ])), cast(PP(*[ptr(g) for g in gres])), cast(IA(*Ks)),
                 cast(IA(*[sp[2] for sp in spec])), cast(IA(*[sp[3] for sp in spec])), cast(IA(*[sp[1] for sp in spec])),
                 ptr(gx0), None, None, st)
            Xs = [x0] + list(Ys[:nl - 1])
            if _deferred is not None and ctx.leaf:
                # the weight-gradient GEMMs of the whole backward pass run as one launch at flush
                gwbs = [_deferred.add_wgrad(GZ[l], Xs[l], Ks[l], N) for l in range(nl)]
            else:
                nb = _hip.query('dig3d_chain_wgrad_workers', M, nl)
                parts = [torch.empty(nb * (N * K + N), dtype=torch.float32, device=dev) for K in Ks]
                gwbs = [torch.empty(N * K + N, dtype=torch.float32, device=dev) for K in Ks]
                call('dig3d_chain_wgrad', nl, cast(PP(*[ptr(g) for g in GZ])), cast(PP(*[ptr(x) for x in Xs])), cast(IA(*Ks)),
                     M, cast(PP(*[ptr(t) for t in parts])), cast(PP(*[ptr(t) for t in gwbs])), 1, st)
            grads = []
            for l in range(nl):
                grads += [gwbs[l][:N * Ks[l]].view(N, Ks[l]), gwbs[l][N * Ks[l]:] if ctx.has[l][0] else None, gres[l]]
            return (gx0, None, None) + tuple(grads)
        # layer-by-layer route (the fused one is checked against it): which earlier layer supplied the saved tile each
        # layer adds (res == 2)
        src, last_saved = [None] * nl, None
        for l in range(nl):
            if spec[l][2] == 2:
                src[l] = last_saved
            if spec[l][3]:
                last_saved = l
        gacc = [None] * nl
        gacc[nl - 1] = _f32c(gout)
        grads = [None] * (3 * nl)
        gx0 = None
        for l in range(nl - 1, -1, -1):
            g = gacc[l]
            K, act, res, _ = spec[l]
            if res == 1:
                grads[3 * l + 2] = g
            elif res == 2:
                s_ = src[l]
                gacc[s_] = g if gacc[s_] is None else gacc[s_] + g
            X = x0 if l == 0 else Ys[l - 1]
            gx = torch.empty(M, K, dtype=torch.float32, device=dev)
            nb = _hip.query('dig3d_linear_wgrad_blocks', M)
            part = torch.empty(nb * (N * K + N), dtype=torch.float32, device=dev)
            gwb = torch.empty(N * K + N, dtype=torch.float32, device=dev)
            z = Zs[l] if act != ACT_NONE else None
            # the gradient already waiting on this layer's input (from a skip connection) is added in the epilogue
            pend = gacc[l - 1] if l > 0 else None
            now = _reduce_later(part, _hip.query('dig3d_linear_bwd_workers', M, K, N), N * K + N, gwb, ctx.leaf)
            call('dig3d_linear_bwd', ptr(g), ptr(z), ptr(Ws[l]), ptr(X), M, K, N, act, ptr(gx), ptr(pend), ptr(part),
                 ptr(gwb), now, st)
            grads[3 * l] = gwb[:N * K].view(N, K)
            if ctx.has[l][0]:
                grads[3 * l + 1] = gwb[N * K:]
            if l == 0:
                gx0 = gx
            else:
                gacc[l - 1] = gx
        return (gx0, None, None) + tuple(grads)


class _Front(Function):
    """The front of an interaction block (spherenet.py:150-163, dimenetpp.py:130-145) as one launch per pass:
        x_ji = swish(lin_ji(x1)),   xd = swish(lin_down(swish(lin_kj(x1)) * rb))
    -> (x_ji, xd, x1', x1''): the last two are aliases of x1 for its OTHER consumers (the skip connection of the layer
    chain, the readout): their gradients come back as separate arguments of ``backward`` and are added inside the kernel
    (csrc/chain.hip:k_front_bwd) instead of by framework additions.  Weight gradients of the three layers: one launch."""

    @staticmethod
    def forward(ctx, x1, rb, Wji, bji, Wkj, bkj, Wd, packed):
        x1, rb = _f32c(x1), _f32c(rb)
        M, ND = x1.size(0), Wd.size(0)
        dev = x1.device
        if packed is None:
            packed = pack_weights([Wji, Wkj, Wd])
        Zji, Xji, Zkj, T = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(4))
        Zd, Xd = (torch.empty(M, ND, dtype=torch.float32, device=dev) for _ in range(2))
        call('dig3d_front_fwd', ptr(x1), M, ptr(packed[0]), ptr(bji), ptr(bkj), ptr(rb), ptr(Zji), ptr(Xji), ptr(Zkj),
             ptr(T), ptr(Zd), ptr(Xd), ND, _stream())
        ctx.save_for_backward(x1, rb, Zji, Zkj, T, Zd, packed)
        ctx.ND = ND
        ctx.has_bias = (bji is not None, bkj is not None)
        ctx.leaf = _all_leaf((Wji, bji, Wkj, bkj, Wd))
        return Xji, Xd, x1.view_as(x1), x1.view_as(x1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gxji, gxd, ga0, ga1):
        x1, rb, Zji, Zkj, T, Zd, packed = ctx.saved_tensors
        M, ND = x1.size(0), ctx.ND
        dev = x1.device
        st = _stream()
        gxji = _f32c(gxji) if gxji is not None else torch.zeros_like(x1)
        gxd = _f32c(gxd) if gxd is not None else torch.zeros_like(Zd)
        ga0 = _f32c(ga0) if ga0 is not None else None
        ga1 = _f32c(ga1) if ga1 is not None else None
        GZji, GZkj, grb, gx1 = (torch.empty(M, 128, dtype=torch.float32, device=dev) for _ in range(4))
        GZd = torch.empty(M, ND, dtype=torch.float32, device=dev)
        call('dig3d_front_bwd', M, ptr(packed[1]), ptr(Zd), ptr(Zkj), ptr(Zji), ptr(rb), ptr(gxd), ptr(gxji), ptr(ga0),
             ptr(ga1), ptr(GZd), ptr(GZkj), ptr(GZji), ptr(grb), ptr(gx1), ND, None, None, None, None, st)
        IA, PP = ctypes.c_int * 3, ctypes.c_void_p * 3
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        Ns = (128, 128, ND)
        GZs, Xs = (GZji, GZkj, GZd), (x1, x1, T)
        if _deferred is not None and ctx.leaf:
            gwbs = [_deferred.add_wgrad(GZs[l], Xs[l], 128, Ns[l]) for l in range(3)]
        else:
            nb = _hip.query('dig3d_chain_wgrad_workers', M, 3)
            parts = [torch.empty(nb * (n * 128 + n), dtype=torch.float32, device=dev) for n in Ns]
            gwbs = [torch.empty(n * 128 + n, dtype=torch.float32, device=dev) for n in Ns]
            call('dig3d_chain_wgrad_n', 3, cast(PP(*[ptr(t) for t in GZs])), cast(PP(*[ptr(t) for t in Xs])),
                 cast(IA(128, 128, 128)), cast(IA(*Ns)), M, cast(PP(*[ptr(t) for t in parts])),
                 cast(PP(*[ptr(t) for t in gwbs])), 1, st)
        gW = [gwbs[l][:Ns[l] * 128].view(Ns[l], 128) for l in range(3)]
        gb = [gwbs[l][Ns[l] * 128:] for l in range(2)]
        return (gx1, grb, gW[0], gb[0] if ctx.has_bias[0] else None, gW[1], gb[1] if ctx.has_bias[1] else None, gW[2],
                None)


def front_supported(x1, rb, lin_ji, lin_kj, lin_down):
    """hidden_channels = 128 (the reference default), int_emb_size a multiple of 16 up to 128, bias-free lin_down,
    float32 on the GPU, first-order gradients only."""
    nd = lin_down.out_features
    return (not _twice_differentiable and x1.is_cuda and x1.dtype == torch.float32 and x1.dim() == 2 and x1.size(0) > 0
            and x1.size(1) == 128 and rb.shape == x1.shape and rb.dtype == torch.float32
            and lin_ji.weight.shape == (128, 128) and lin_kj.weight.shape == (128, 128)
            and lin_down.in_features == 128 and lin_down.bias is None and nd % 16 == 0 and 16 <= nd <= 128)


def front(x1, rb, lin_ji, lin_kj, lin_down, packed=None):
    """-> (x_ji, xd, x1 for the chain's skip connection, x1 for the readout) — see ``_Front``."""
    return _Front.apply(x1, rb, lin_ji.weight, lin_ji.bias, lin_kj.weight, lin_kj.bias, lin_down.weight, packed)


# route selectors the tests flip to compare a fused kernel with the route it replaced (not configuration: defaults = the
# measured winners of rounds 2-3, docs/history/DESIGN_rounds_1_to_5.md §6)
_chain_bwd_fused = True
linear_defer_min = 1 << 29      # M N K from which a dense layer's weight gradient joins the pass's one deferred launch
force_wgrad_deferred = True     # energy_and_force: the chain / front weight gradients of the final pass join that launch too
#                                 (False: one k_chain_wgrad launch per chain, front and contribution — 16 per DimeNet++ step)
_wide_chain = True
_embed_kernel = True


def chain_supported(x0, layers):
    """layers: list of (weight, bias, act, res_kind, res_tensor, save).  The fused chain covers <= 8 layers of 128
    outputs, K_0 <= 128 (multiple of 8) and K_l = 128 afterwards, swish or no activation (the only ones the reference's
    interaction blocks use, spherenet.py:34-50,172-182), float32 on the GPU, first-order gradients only."""
    if _twice_differentiable or not (1 <= len(layers) <= 8) or not x0.is_cuda or x0.dtype != torch.float32:
        return False
    if x0.dim() != 2 or x0.size(0) == 0:
        return False
    for l, (w, b, act, res, rt, save) in enumerate(layers):
        K = w.size(1)
        if w.size(0) != 128 or K > 128 or K % 8 or (l > 0 and K != 128) or act not in (ACT_NONE, ACT_SWISH):
            return False
    return layers[0][0].size(1) == x0.size(1)


def chain(x0, layers, packed=None):
    spec = tuple((w.size(1), act, res, int(bool(save))) for (w, b, act, res, rt, save) in layers)
    flat = []
    for (w, b, act, res, rt, save) in layers:
        flat += [w, b, rt if res == 1 else None]
    return _Chain.apply(x0, spec, packed, *flat)


def pack_weights(Ws):
    """[2, L, 16384]: the weights W_l [N_l <= 128, K_l <= 128] re-laid in MFMA operand order, forward ([0]) and backward
    ([1]) formats — csrc/chain.hip:k_chain_pack, ONE launch for up to 64 layers.  A re-layout of the current values, not
    part of the autograd graph: the chain / front Functions differentiate w.r.t. the original tensors."""
    L = len(Ws)
    Ws = [_f32c(w.detach()) for w in Ws]
    packed = torch.empty(2, L, 16384, dtype=torch.float32, device=Ws[0].device)
    PP, IA = ctypes.c_void_p * L, ctypes.c_int * L
    cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
    call('dig3d_chain_pack', L, cast(PP(*[ptr(w) for w in Ws])), cast(IA(*[w.size(1) for w in Ws])),
         cast(IA(*[w.size(0) for w in Ws])), ptr(packed[0]), ptr(packed[1]), _stream())
    return packed


def packable(Ws):
    return (1 <= len(Ws) <= 64 and all(w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.size(0) % 16 == 0
                                       and 16 <= w.size(0) <= 128 and w.size(1) % 8 == 0 and 8 <= w.size(1) <= 128 for w in Ws))


# --- twice-differentiable route: three matmul forms on the MFMA kernels, closed under differentiation -------------
#   NT: C[M,N] = A[M,K] B[N,K]^T    NN: C[M,K] = A[M,N] B[N,K]    TN: C[N,K] = A[M,N]^T B[M,K]
# d(NT) = (NN, TN), d(NN) = (NT, TN), d(TN) = (NT, NN): any order of differentiation stays on these kernels, which is
# what the energy_and_force path needs (run.py:126: force = -dE/dpos with create_graph, then loss.backward()).
class _MatmulNT(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        ctx.save_for_backward(a, b)
        (M, K), N = a.shape, b.size(0)
        c = torch.empty(M, N, dtype=torch.float32, device=a.device)
        call('dig3d_linear_fwd', ptr(a), ptr(b), None, None, M, K, N, ACT_NONE, ptr(c), None, _stream())
        return c

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (matmul_nn(g, b) if ctx.needs_input_grad[0] else None,
                matmul_tn(g, a) if ctx.needs_input_grad[1] else None)


class _MatmulNN(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        ctx.save_for_backward(a, b)
        (M, N), K = a.shape, b.size(1)
        c = torch.empty(M, K, dtype=torch.float32, device=a.device)
        call('dig3d_linear_bwd_input', ptr(a), None, ptr(b), M, K, N, ACT_NONE, ptr(c), None, _stream())
        return c

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (matmul_nt(g, b) if ctx.needs_input_grad[0] else None,
                matmul_tn(a, g) if ctx.needs_input_grad[1] else None)


class _MatmulTN(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        ctx.save_for_backward(a, b)
        (M, N), K = a.shape, b.size(1)
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        part = torch.empty(nb * (N * K + N), dtype=torch.float32, device=a.device)
        cwb = torch.empty(N * K + N, dtype=torch.float32, device=a.device)
        call('dig3d_linear_bwd_weight', ptr(a), None, ptr(b), M, K, N, ACT_NONE, ptr(part), ptr(cwb), 1, _stream())
        return cwb[:N * K].view(N, K)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (matmul_nt(b, g) if ctx.needs_input_grad[0] else None,
                matmul_nn(a, g) if ctx.needs_input_grad[1] else None)


def _mm_check(*ts):
    if not all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.size(0) > 0 for t in ts):
        raise _hip.Dig3dError('dig_amd.ops.matmul_*: float32 GPU matrices with at least one row expected '
                              f'(got {[(tuple(t.shape), t.dtype, t.device.type) for t in ts]}); the engine has no framework fallback')


def _up8(n):
    return (n + 7) & ~7


class _Pad2d(Function):
    """dst[rows, cols] = src zero-padded / sliced (csrc/readout.hip:k_pad2d).  Linear; the adjoint is the same map with the
    shapes exchanged, so the Function is its own backward and differentiable any number of times."""

    @staticmethod
    def forward(ctx, src, rows, cols):
        ctx.shape = tuple(src.shape)
        src = _f32c(src)
        out = torch.empty(rows, cols, dtype=torch.float32, device=src.device)
        call('dig3d_pad2d', ptr(src) if src.numel() else None, src.size(0), src.size(1), ptr(out), rows, cols, _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return _Pad2d.apply(g, *ctx.shape), None, None


def pad2d(src, rows, cols):
    """[rows, cols] from a 2-D float32 tensor: zero padding where it grows, slicing where it shrinks"""
    if src.size(0) == rows and src.size(1) == cols:
        return src
    return _Pad2d.apply(src, rows, cols)


def matmul_nt(a, b):
    """a[M,K] @ b[N,K]^T"""
    _mm_check(a, b)
    N = b.size(0)
    if N & 7:                                   # output width to the next multiple of 8 (zero rows of b), sliced back
        return pad2d(_MatmulNT.apply(a, pad2d(b, _up8(N), b.size(1))), a.size(0), N)
    return _MatmulNT.apply(a, b)


def matmul_nn(a, b):
    """a[M,N] @ b[N,K]"""
    _mm_check(a, b)
    N = a.size(1)
    if N & 7:                                   # the reduction dimension: zero columns of a, zero rows of b
        return _MatmulNN.apply(pad2d(a, a.size(0), _up8(N)), pad2d(b, _up8(N), b.size(1)))
    return _MatmulNN.apply(a, b)


def matmul_tn(a, b):
    """a[M,N]^T @ b[M,K]"""
    _mm_check(a, b)
    N = a.size(1)
    if N & 7:
        return pad2d(_MatmulTN.apply(pad2d(a, a.size(0), _up8(N)), b), N, b.size(1))
    return _MatmulTN.apply(a, b)


def linear_rowscale(x, weight, bias, rowscale):
    """``F.linear(x, weight, bias) * rowscale.view(-1, 1)`` for a row factor that needs no gradient (SchNet's cosine
    cutoff on the generated filters, schnet.py:31-33): one MFMA launch, and a backward without the extra multiplies."""
    N = weight.size(0)
    if (x.is_cuda and not _twice_differentiable and x.dim() == 2 and x.dtype == weight.dtype == torch.float32
            and (N & 7) == 0 and x.size(0) > 0 and x.size(1) > 16 and not rowscale.requires_grad
            and rowscale.numel() == x.size(0) and rowscale.dtype == torch.float32):
        return _LinearAct.apply(x, weight, bias, None, ACT_NONE, rowscale.reshape(-1))
    return linear(x, weight, bias) * rowscale.view(-1, 1)


_warned_cast = False


def linear(x, weight, bias=None, act=ACT_NONE, res=None):
    """act(F.linear(x, weight, bias)) (+ res) — the hidden-channel layers of every interaction block."""
    K, N = weight.size(1), weight.size(0)
    if not x.is_cuda:
        raise _hip.Dig3dError('dig_amd op received a CPU tensor; the engine has no CPU fallback')
    if x.dim() != 2:                            # F.linear over leading dimensions: rows are rows
        lead = x.shape[:-1]
        y = linear(x.reshape(-1, K), weight, bias, act, None if res is None else res.reshape(-1, N))
        return y.reshape(*lead, N)
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        # model.double() / half inputs (the reference's nn.Linear takes any dtype): the engine computes in float32 — the
        # operands are cast (differentiably), the result goes back to the input's dtype; said once (this raised)
        if not (x.is_floating_point() and weight.is_floating_point()):
            raise RuntimeError(f'linear: expected floating-point tensors, got {x.dtype} x {weight.dtype}')
        global _warned_cast
        if not _warned_cast:
            import warnings
            warnings.warn(f'dig_amd.ops.linear computes in float32: {x.dtype} x {weight.dtype} operands are cast')
            _warned_cast = True
        y = linear(x.float(), weight.float(), None if bias is None else bias.float(), act, None if res is None else res.float())
        return y.to(x.dtype)
    if x.size(0) == 0:                          # an empty batch: an empty result that still reaches the weights' gradients
        y = x.new_zeros(0, N) + x.sum() * 0 + weight.sum() * 0
        return y if bias is None else y + bias.sum() * 0
    small_head = (not _twice_differentiable and 1 <= N <= 8 and act == ACT_NONE and res is None)
    if (N & 7) and not small_head:
        # a width that is not a multiple of 8 (spherenet.py:253-259 accepts any hidden_channels / int_emb_size; the 256 -> 1
        # heads of the force route): the weight's rows, the bias and the residual are zero-padded to the next multiple of 8
        # (act(0) = 0 for swish and shifted softplus, the extra columns stay zero), the SAME MFMA kernels run, the result is
        # sliced back.  Three to five small copies per layer instead of a library GEMM + separate activation kernels; pad2d
        # is closed under differentiation, so this also serves the energy_and_force route.
        N8, M = _up8(N), x.size(0)
        w8 = pad2d(weight, N8, K)
        b8 = None if bias is None else pad2d(bias.reshape(1, N), 1, N8).reshape(N8)
        r8 = None if res is None else pad2d(res, M, N8)
        return pad2d(linear(x, w8, b8, act, r8), M, N)
    if _twice_differentiable and act in (ACT_NONE, ACT_SWISH, ACT_SSP):
        # one MFMA kernel forward; backward and double backward on MFMA + two elementwise kernels (dig_amd/diffops.py)
        from . import diffops
        return diffops.linear2(x, weight, bias, act, res)
    if small_head:
        # a head with a few outputs (lin_out: 256 -> 1, comenet.py:286; SchNet's lin2, schnet.py:236): row dot products
        # (csrc/readout.hip) — the library runs these as a GEMV, an outer product and a split-K GEMM of 56 us
        return _GroupedSmallN.apply(1, x, weight, bias)[0]
    return _LinearAct.apply(x, weight, bias, res, act)


# ---------------------------------------------------------------------------------------------------
# grouped output blocks (csrc/readout.hip, csrc/dense.hip grouped kernels): the L + 1 ``update_v`` / ``update_u`` blocks
# of a SphereNet / DimeNet++ forward (spherenet.py:185-225) are independent of each other, so every stage of all of
# them is ONE launch instead of L + 1.
# ---------------------------------------------------------------------------------------------------
def _ptrs(ts):
    arr = (ctypes.c_void_p * len(ts))(*[ptr(t) for t in ts])
    return ctypes.cast(arr, ctypes.c_void_p), arr


class _GroupedSegSum(Function):
    """outs[g] = segment_sum(As[g] * Bs[g], seg) for a sorted segmentation shared by all groups: the product
    e2 = lin_rbf(rbf) * e1 (spherenet.py:90,182) is formed while summing, never written."""

    @staticmethod
    def forward(ctx, seg, G, *tensors):
        As = [_f32c(x) for x in tensors[:G]]
        Bs = [_f32c(x) for x in tensors[G:2 * G]]
        C = As[0].size(1)
        outs = [torch.empty(seg.S, C, dtype=torch.float32, device=As[0].device) for _ in range(G)]
        pa, k1 = _ptrs(As)
        pb, k2 = _ptrs(Bs)
        po, k3 = _ptrs(outs)
        call('dig3d_segment_sum_grouped', G, pa, pb, ptr(seg.kptr), seg.S, C, po, _stream())
        ctx.seg, ctx.G = seg, G
        ctx.save_for_backward(*As, *Bs)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        seg, G = ctx.seg, ctx.G
        sv = ctx.saved_tensors
        As, Bs = sv[:G], sv[G:]
        gs = [_f32c(g) for g in gs]
        C, M = gs[0].size(1), seg.key.numel()
        dev = gs[0].device
        gA = [torch.empty(M, C, dtype=torch.float32, device=dev) for _ in range(G)]
        gB = [torch.empty(M, C, dtype=torch.float32, device=dev) for _ in range(G)]
        pi, k1 = _ptrs(gs)
        po, k2 = _ptrs(gA)
        pm, k3 = _ptrs(Bs)
        po2, k4 = _ptrs(gB)
        pm2, k5 = _ptrs(As)
        call('dig3d_gather_grouped', G, pi, ptr(seg.key), M, C, po, pm, po2, pm2, ptr(seg.cnt), _stream())
        return (None, None) + tuple(gA) + tuple(gB)


class _GroupedLinear(Function):
    """ys[g] = act(xs[g] Ws[g]^T + bs[g]) for G layers of one shape: one MFMA launch forward, one backward."""

    @staticmethod
    def forward(ctx, act, G, *tensors):
        xs = [_f32c(t) for t in tensors[:G]]
        Ws = [_f32c(t) for t in tensors[G:2 * G]]
        bs = list(tensors[2 * G:3 * G])
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        ys = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)]
        zs = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)] if act != ACT_NONE else [None] * G
        px, k1 = _ptrs(xs)
        pw, k2 = _ptrs(Ws)
        pb, k3 = _ptrs(bs)
        py, k4 = _ptrs(ys)
        pz, k5 = _ptrs(zs)
        call('dig3d_linear_fwd_grouped', G, px, pw, pb, None, M, K, N, act, py, pz, _stream())
        ctx.act, ctx.G, ctx.has_bias = act, G, [b is not None for b in bs]
        ctx.leaf = _all_leaf(Ws) and _all_leaf(bs)
        ctx.save_for_backward(*xs, *Ws, *[z if z is not None else xs[0].new_empty(0) for z in zs])
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        G, act = ctx.G, ctx.act
        sv = ctx.saved_tensors
        xs, Ws, zs = sv[:G], sv[G:2 * G], sv[2 * G:3 * G]
        gys = [_f32c(g) for g in gys]
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        stride = N * K + N
        gxs = [torch.empty(M, K, dtype=torch.float32, device=dev) for _ in range(G)]
        pg, k1 = _ptrs(gys)
        pz, k2 = _ptrs([z if act != ACT_NONE else None for z in zs])
        pw, k3 = _ptrs(Ws)
        pgx, k5 = _ptrs(gxs)
        # (splitting these into a grouped input-gradient launch + the backward pass's single weight-gradient launch was
        # measured and dropped: at ~600 rows the input-gradient launch alone costs what the merged one does — 20 vs 22 us)
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        parts = [torch.empty(nb * stride, dtype=torch.float32, device=dev) for _ in range(G)]
        gwbs = [torch.empty(stride, dtype=torch.float32, device=dev) for _ in range(G)]
        now = [_reduce_later(parts[g], nb, stride, gwbs[g], ctx.leaf) for g in range(G)][0]
        px, k4 = _ptrs(xs)
        pp, k6 = _ptrs(parts)
        pgw, k7 = _ptrs(gwbs)
        call('dig3d_linear_bwd_grouped', G, pg, pz, pw, px, M, K, N, act, pgx, None, pp, pgw, now, None, _stream())
        gws = [w[:N * K].view(N, K) for w in gwbs]
        gbs = [(w[N * K:] if hb else None) for w, hb in zip(gwbs, ctx.has_bias)]
        return (None, None) + tuple(gxs) + tuple(gws) + tuple(gbs)


class _GroupedLinearRes(Function):
    """ys[g] = act(xs[g] Ws[g]^T + bs[g]) + ress[g] for G layers of one shape (residuals optional): one MFMA launch forward,
    one backward — for the pairs of independent layers of a ComENet block at a few hundred atoms (comenet.py:130-133,
    199-208: conv1 / conv2 ``lin_root``, ``lin_rel`` + root, ``lin1`` / ``lin2``), where every launch is ~20 us of latency
    whatever its size."""

    @staticmethod
    def forward(ctx, act, G, *tensors):
        xs = [_f32c(t) for t in tensors[:G]]
        Ws = [_f32c(t) for t in tensors[G:2 * G]]
        bs = list(tensors[2 * G:3 * G])
        rs = [(_f32c(t) if t is not None else None) for t in tensors[3 * G:4 * G]]
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        ys = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)]
        zs = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(G)] if act != ACT_NONE else [None] * G
        px, k1 = _ptrs(xs)
        pw, k2 = _ptrs(Ws)
        pb, k3 = _ptrs(bs)
        pr, k6 = _ptrs(rs)
        py, k4 = _ptrs(ys)
        pz, k5 = _ptrs(zs)
        call('dig3d_linear_fwd_grouped', G, px, pw, pb, pr, M, K, N, act, py, pz, _stream())
        if act == _ACT_ROWSCALE:
            # the res slot carried a row factor (no gradient), Z its broadcast: the layer's "derivative tensor" (as _LinearAct)
            act, rs = _ACT_DERIV, [None] * G
        ctx.act, ctx.G, ctx.has_bias, ctx.has_res = act, G, [b is not None for b in bs], [r is not None for r in rs]
        ctx.leaf = _all_leaf(Ws) and _all_leaf(bs)
        ctx.save_for_backward(*xs, *Ws, *[z if z is not None else xs[0].new_empty(0) for z in zs])
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        G, act = ctx.G, ctx.act
        sv = ctx.saved_tensors
        xs, Ws, zs = sv[:G], sv[G:2 * G], sv[2 * G:3 * G]
        gys = [_f32c(g) for g in gys]
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        stride = N * K + N
        # (no input of the group needs a gradient — SchNet's filter network on the Gaussian smearing: the input-gradient tiles
        # are not launched)
        want_x = any(ctx.needs_input_grad[2 + g] for g in range(G))
        gxs = [torch.empty(M, K, dtype=torch.float32, device=dev) if want_x else None for _ in range(G)]
        pg, k1 = _ptrs(gys)
        pz, k2 = _ptrs([z if act != ACT_NONE else None for z in zs])
        pw, k3 = _ptrs(Ws)
        pgx, k5 = _ptrs(gxs) if want_x else (None, None)
        nb = _hip.query('dig3d_linear_wgrad_blocks', M)
        parts = [torch.empty(nb * stride, dtype=torch.float32, device=dev) for _ in range(G)]
        gwbs = [torch.empty(stride, dtype=torch.float32, device=dev) for _ in range(G)]
        now = [_reduce_later(parts[g], nb, stride, gwbs[g], ctx.leaf) for g in range(G)][0]
        px, k4 = _ptrs(xs)
        pp, k6 = _ptrs(parts)
        pgw, k7 = _ptrs(gwbs)
        call('dig3d_linear_bwd_grouped', G, pg, pz, pw, px, M, K, N, act, pgx, None, pp, pgw, now, None, _stream())
        gws = [w[:N * K].view(N, K) for w in gwbs]
        gbs = [(w[N * K:] if hb else None) for w, hb in zip(gwbs, ctx.has_bias)]
        grs = [(g if hr else None) for g, hr in zip(gys, ctx.has_res)]          # y = ... + res: the residual's gradient is gy
        return (None, None) + tuple(gxs) + tuple(gws) + tuple(gbs) + tuple(grs)


def grouped_linear_supported(xs, Ws):
    """G <= 8 float32 GPU layers of ONE shape with an output width the MFMA kernels take (multiple of 8), once differentiable"""
    if _twice_differentiable or not (1 <= len(xs) <= 8) or len(xs) != len(Ws):
        return False
    M, K = xs[0].shape if xs[0].dim() == 2 else (0, 0)
    N = Ws[0].size(0)
    return (M > 0 and (N & 7) == 0 and all(x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (M, K) for x in xs)
            and all(w.dtype == torch.float32 and tuple(w.shape) == (N, K) for w in Ws))


_ACT_DERIV, _ACT_ROWSCALE = 3, 7        # csrc/dense_common.h


def grouped_linear_rowscale(xs, Ws, bs, rowscale):
    """[(x_g W_g^T + b_g) * rowscale.view(-1, 1)] for G same-shape layers sharing one row factor that needs no gradient
    (SchNet's cosine cutoff on the generated filters of every block, schnet.py:31-33) — one launch per pass."""
    G = len(xs)
    rs = _f32c(rowscale).reshape(-1)
    return list(_GroupedLinearRes.apply(_ACT_ROWSCALE, G, *xs, *Ws, *bs, *([rs] * G)))


def grouped_linear(xs, Ws, bs, act=ACT_NONE, ress=None):
    """[act(x_g W_g^T + b_g) (+ res_g)] for G same-shape layers — one launch per pass (``grouped_linear_supported``)."""
    G = len(xs)
    ress = list(ress) if ress is not None else [None] * G
    return list(_GroupedLinearRes.apply(act, G, *xs, *Ws, *bs, *ress))


class _WideChain(Function):
    """G independent chains of nl <= 4 layers with 256 outputs, one launch per pass (csrc/wide.hip):
        Y_l = res_l * Y_{l-1} + act_l(Y_{l-1} W_l^T + b_l),   K_0 in {128, 256}, K_l = 256 afterwards
    — the output blocks of all interaction layers of SphereNet / DimeNet++ (lin_up + lins, spherenet.py:185-216; G = L + 1
    groups) and ComENet's residual layers (comenet.py:209-210; G = 1, res = 1).  Backward: the input-gradient recursion
    of all layers and groups in one launch; the weight gradients join the step's single weight-gradient launch
    (``deferred_reductions.add_wgrad``).  tensors = xs[G], then per group and layer (weight, bias)."""

    @staticmethod
    def forward(ctx, G, spec, *tensors):
        nl = len(spec)
        xs = [_f32c(t) for t in tensors[:G]]
        rest = tensors[G:]
        Ws = [_f32c(rest[2 * i]) for i in range(G * nl)]
        bs = [rest[2 * i + 1] for i in range(G * nl)]
        M, K0 = xs[0].shape
        dev = xs[0].device
        n = G * nl
        Ks = [K0 if (i % nl) == 0 else 256 for i in range(n)]
        packed = torch.empty(n, 2, 65536, dtype=torch.float32, device=dev)     # forward / backward operand order
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        PPn, IAn = ctypes.c_void_p * n, ctypes.c_int * n
        call('dig3d_wide_pack', n, cast(PPn(*[ptr(w) for w in Ws])), cast(IAn(*Ks)),
             cast(PPn(*[packed[i, 0].data_ptr() for i in range(n)])), cast(PPn(*[packed[i, 1].data_ptr() for i in range(n)])),
             _stream())
        Zs = [torch.empty(M, 256, dtype=torch.float32, device=dev) if spec[i % nl][0] != ACT_NONE else None for i in range(n)]
        Ys = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        IAl, PPg = ctypes.c_int * nl, ctypes.c_void_p * G
        call('dig3d_wide_fwd', G, nl, M, K0, cast(PPg(*[ptr(x) for x in xs])),
             cast(PPn(*[packed[i, 0].data_ptr() for i in range(n)])), cast(PPn(*[ptr(b) for b in bs])),
             cast(PPn(*[ptr(z) for z in Zs])), cast(PPn(*[ptr(y) for y in Ys])),
             cast(IAl(*[1 if sp[0] == ACT_SWISH else 0 for sp in spec])), cast(IAl(*[int(sp[1]) for sp in spec])), _stream())
        ctx.G, ctx.spec, ctx.K0 = G, spec, K0
        ctx.has_bias = [b is not None for b in bs]
        ctx.leaf = _all_leaf(Ws) and _all_leaf(bs)
        ctx.save_for_backward(packed, *xs, *[z if z is not None else xs[0].new_empty(0) for z in Zs],
                              *[Ys[i] for i in range(n) if (i % nl) != nl - 1])
        return tuple(Ys[g * nl + nl - 1] for g in range(G))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        G, spec, K0 = ctx.G, ctx.spec, ctx.K0
        nl = len(spec)
        n = G * nl
        sv = ctx.saved_tensors
        packed, xs, Zs, inner = sv[0], sv[1:1 + G], sv[1 + G:1 + G + n], list(sv[1 + G + n:])
        M = xs[0].size(0)
        dev = xs[0].device
        # layer inputs: X_0 = the chain input, X_l = Y_{l-1}
        Xin = []
        it = iter(inner)
        for g in range(G):
            Xin.append(xs[g])
            for l in range(nl - 1):
                Xin.append(next(it))
        gys = [_f32c(gy) if gy is not None else torch.zeros(M, 256, dtype=torch.float32, device=dev) for gy in gys]
        GZ = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(n)]
        gx0 = [torch.empty(M, K0, dtype=torch.float32, device=dev) for _ in range(G)]
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        PPn, PPg, IAl = ctypes.c_void_p * n, ctypes.c_void_p * G, ctypes.c_int * nl
        zs = [Zs[i] if spec[i % nl][0] != ACT_NONE else None for i in range(n)]
        call('dig3d_wide_bwd', G, nl, M, K0, cast(PPg(*[ptr(t) for t in gys])),
             cast(PPn(*[packed[i, 1].data_ptr() for i in range(n)])), cast(PPn(*[ptr(z) for z in zs])),
             cast(PPn(*[ptr(t) for t in GZ])), cast(PPg(*[ptr(t) for t in gx0])), None,
             cast(IAl(*[1 if sp[0] == ACT_SWISH else 0 for sp in spec])), cast(IAl(*[int(sp[1]) for sp in spec])), None, None,
             _stream())
        Ks = [K0 if (i % nl) == 0 else 256 for i in range(n)]
        if _deferred is not None and ctx.leaf:
            gwbs = [_deferred.add_wgrad(GZ[i], Xin[i], Ks[i], 256) for i in range(n)]
        else:                                  # stand-alone backward: the same launch, flushed here
            with deferred_reductions() as red:
                gwbs = [red.add_wgrad(GZ[i], Xin[i], Ks[i], 256) for i in range(n)]
            red.flush()
        grads = []
        for i in range(n):
            grads += [gwbs[i][:256 * Ks[i]].view(256, Ks[i]), gwbs[i][256 * Ks[i]:] if ctx.has_bias[i] else None]
        return (None, None) + tuple(gx0) + tuple(grads)


def wide_chain_supported(xs, layers):
    """xs: G inputs of one shape [M, 128 | 256]; layers: per group a list of (weight [256, K], bias, act, res)."""
    if _twice_differentiable or not xs or len(xs) > 8 or not layers or len(layers) != len(xs):
        return False
    nl = len(layers[0])
    M, K0 = xs[0].shape if xs[0].dim() == 2 else (0, 0)
    if not (1 <= nl <= 4) or K0 not in (128, 256) or M == 0:
        return False
    for x, ls in zip(xs, layers):
        if not x.is_cuda or x.dtype != torch.float32 or tuple(x.shape) != (M, K0) or len(ls) != nl:
            return False
        for l, (w, b, act, res) in enumerate(ls):
            if tuple(w.shape) != (256, K0 if l == 0 else 256) or act not in (ACT_NONE, ACT_SWISH):
                return False
            if (act, bool(res)) != (layers[0][l][2], bool(layers[0][l][3])) or (res and w.size(1) != 256):
                return False
    return bool(_hip.query('dig3d_wide_supported', M, K0, nl, len(xs)))


def wide_chain(xs, layers):
    """-> tuple of the G chain outputs [M, 256] (see ``_WideChain``)."""
    spec = tuple((act, int(bool(res))) for (_, _, act, res) in layers[0])
    flat = []
    for ls in layers:
        for (w, b, _, _) in ls:
            flat += [w, b]
    return _WideChain.apply(len(xs), spec, *xs, *flat)


class _GroupedSmallN(Function):
    """ys[g] = xs[g] Ws[g]^T (+ bs[g]) with <= 8 outputs: the ``lin`` heads of the output blocks as row dot products."""

    @staticmethod
    def forward(ctx, G, *tensors):
        xs = [_f32c(t) for t in tensors[:G]]
        Ws = [_f32c(t) for t in tensors[G:2 * G]]
        bs = list(tensors[2 * G:3 * G])
        M, K = xs[0].shape
        N = Ws[0].size(0)
        ys = [torch.empty(M, N, dtype=torch.float32, device=xs[0].device) for _ in range(G)]
        px, k1 = _ptrs(xs)
        pw, k2 = _ptrs(Ws)
        pb, k3 = _ptrs(bs)
        py, k4 = _ptrs(ys)
        call('dig3d_smalln_fwd_grouped', G, px, pw, pb, M, K, N, py, _stream())
        ctx.G, ctx.has_bias = G, [b is not None for b in bs]
        ctx.leaf = _all_leaf(Ws) and _all_leaf(bs)
        ctx.save_for_backward(*xs, *Ws)
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        G = ctx.G
        sv = ctx.saved_tensors
        xs, Ws = sv[:G], sv[G:2 * G]
        gys = [_f32c(g) for g in gys]
        M, K = xs[0].shape
        N = Ws[0].size(0)
        dev = xs[0].device
        stride = N * K + N
        nb = _hip.query('dig3d_smalln_blocks', M)
        gxs = [torch.empty(M, K, dtype=torch.float32, device=dev) for _ in range(G)]
        parts = [torch.empty(nb * stride, dtype=torch.float32, device=dev) for _ in range(G)]
        gwbs = [torch.empty(stride, dtype=torch.float32, device=dev) for _ in range(G)]
        pg, k1 = _ptrs(gys)
        pw, k2 = _ptrs(Ws)
        px, k3 = _ptrs(xs)
        pgx, k4 = _ptrs(gxs)
        pp, k5 = _ptrs(parts)
        call('dig3d_smalln_bwd_grouped', G, pg, pw, px, M, K, N, pgx, pp, _stream())
        now = [_reduce_later(parts[g], nb, stride, gwbs[g], ctx.leaf) for g in range(G)][0]
        if now:
            PP, IA, LA = ctypes.c_void_p * G, ctypes.c_int * G, ctypes.c_int64 * G
            cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
            call('dig3d_reduce_many', cast(PP(*[ptr(q) for q in parts])), cast(IA(*[nb] * G)), cast(LA(*[stride] * G)),
                 cast(IA(*[stride] * G)), cast(PP(*[ptr(q) for q in gwbs])), G, _stream())
        gws = [w[:N * K].view(N, K) for w in gwbs]
        gbs = [(w[N * K:] if hb else None) for w, hb in zip(gwbs, ctx.has_bias)]
        return (None,) + tuple(gxs) + tuple(gws) + tuple(gbs)


class _GroupedGraphSum(Function):
    """u = ((0 + scatter(ys[0], batch)) + scatter(ys[1], batch)) + ...  — update_u of every block, reference order."""

    @staticmethod
    def forward(ctx, seg, *ys):
        ys = [_f32c(y) for y in ys]
        G, C = len(ys), ys[0].size(1)
        u = torch.empty(seg.S, C, dtype=torch.float32, device=ys[0].device)
        py, k1 = _ptrs(ys)
        call('dig3d_graph_sum_grouped', G, py, ptr(seg.kptr), seg.S, C, ptr(u), _stream())
        ctx.seg, ctx.G = seg, G
        return u

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gu):
        seg = ctx.seg
        gy = gather_mul_raw(_f32c(gu), seg.key, cnt=seg.cnt)       # the same gradient for every block's y
        return (None,) + (gy,) * ctx.G


class _RadialBundle(Function):
    """every radial-basis projection of a forward in ONE launch (csrc/radial.hip), all their backward passes in one
    more.  ``spec[h]`` = (two_layer, has_bias, act); tensors = per head (Wa, bias-or-None) or (Wa, Wb)."""

    @staticmethod
    def forward(ctx, x, spec, *tensors):
        x = _f32c(x)
        M, K = x.shape
        H = len(spec)
        Wa = [_f32c(tensors[2 * h]) for h in range(H)]
        second = [tensors[2 * h + 1] for h in range(H)]
        Wb = [(_f32c(second[h]) if spec[h][0] else None) for h in range(H)]
        bias = [(second[h] if not spec[h][0] else None) for h in range(H)]
        N = [(Wb[h].size(0) if spec[h][0] else Wa[h].size(0)) for h in range(H)]
        J = [Wa[h].size(0) for h in range(H)]
        act = [spec[h][2] for h in range(H)]
        Y = [torch.empty(M, N[h], dtype=torch.float32, device=x.device) for h in range(H)]
        IA = ctypes.c_int * H
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        pa, k1 = _ptrs(Wa)
        pb, k2 = _ptrs(Wb)
        pbias, k3 = _ptrs(bias)
        py, k4 = _ptrs(Y)
        ctx.ints = (IA(*N), IA(*J), IA(*act), IA(*[int(sp[0]) for sp in spec]))
        call('dig3d_radial_fwd', ptr(x), M, K, H, pa, pb, pbias, cast(ctx.ints[0]), cast(ctx.ints[1]), cast(ctx.ints[2]), py,
             _stream())
        ctx.spec, ctx.N, ctx.J = spec, N, J
        ctx.leaf = _all_leaf(tensors)
        ctx.save_for_backward(x, *Wa, *[w if w is not None else x.new_empty(0) for w in Wb],
                              *[b if b is not None else x.new_empty(0) for b in bias])
        return tuple(Y)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gY):
        spec, N, J = ctx.spec, ctx.N, ctx.J
        H = len(spec)
        sv = ctx.saved_tensors
        x, Wa, Wb, bias = sv[0], sv[1:1 + H], sv[1 + H:1 + 2 * H], sv[1 + 2 * H:1 + 3 * H]
        M, K = x.shape
        dev = x.device
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        stride = _hip.query('dig3d_radial_partial_stride', H, cast(ctx.ints[0]), cast(ctx.ints[1]), cast(ctx.ints[3]), K)
        nb = _hip.query('dig3d_radial_blocks', M, H)
        gY = [(_f32c(g) if g is not None else None) for g in gY]
        gX = torch.empty(M, K, dtype=torch.float32, device=dev)
        part = torch.empty(nb * stride, dtype=torch.float32, device=dev)
        gall = torch.empty(stride, dtype=torch.float32, device=dev)
        pa, k1 = _ptrs(Wa)
        pb, k2 = _ptrs([(Wb[h] if spec[h][0] else None) for h in range(H)])
        pbias, k3 = _ptrs([(bias[h] if (not spec[h][0] and spec[h][1]) else None) for h in range(H)])
        pg, k4 = _ptrs(gY)
        G = _hip.query('dig3d_radial_bwd_groups', H)          # one gX slice per head
        work = torch.empty(G * M * K, dtype=torch.float32, device=dev) if G > 1 else None
        call('dig3d_radial_bwd', ptr(x), M, K, H, pa, pb, pbias, cast(ctx.ints[0]), cast(ctx.ints[1]), cast(ctx.ints[2]), pg,
             ptr(gX), ptr(part), ptr(work), _stream())
        if _reduce_later(part, nb, stride, gall, ctx.leaf):
            PP, IA, LA = ctypes.c_void_p * 1, ctypes.c_int * 1, ctypes.c_int64 * 1
            call('dig3d_reduce_many', cast(PP(ptr(part))), cast(IA(nb)), cast(LA(stride)), cast(IA(stride)),
                 cast(PP(ptr(gall))), 1, _stream())
        grads, off = [], 0
        for h in range(H):
            if spec[h][0]:
                ga = gall[off:off + J[h] * K].view(J[h], K)
                gb = gall[off + J[h] * K:off + J[h] * K + N[h] * J[h]].view(N[h], J[h])
                off += J[h] * K + N[h] * J[h]
            else:
                ga = gall[off:off + N[h] * K].view(N[h], K)
                gb = gall[off + N[h] * K:off + N[h] * K + N[h]] if spec[h][1] else None
                off += N[h] * K + N[h]
            grads += [ga, gb]
        return (gX, None) + tuple(grads)


def radial_bundle_supported(K, heads):
    """heads: [(N, J or None)].  K = num_radial <= 8, J = basis_emb_size <= 8, N % 4 == 0, 8 <= N <= 256, <= 16 heads."""
    return (not _twice_differentiable and 1 <= K <= 8 and 1 <= len(heads) <= 16
            and all(8 <= n <= 256 and n % 4 == 0 and (j is None or 1 <= j <= 8) for n, j in heads))


def radial_bundle(x, heads):
    """heads: list of ('single', weight, bias_or_None, act) / ('two', W1, W2) -> list of outputs [M, N_h]."""
    spec, flat = [], []
    for hd in heads:
        if hd[0] == 'two':
            spec.append((True, False, ACT_NONE))
            flat += [hd[1], hd[2]]
        else:
            spec.append((False, hd[2] is not None, hd[3]))
            flat += [hd[1], hd[2]]
    return list(_RadialBundle.apply(x, tuple(spec), *flat))


def graph_sum_group(ys, seg_batch):
    """((0 + scatter(ys[0], batch)) + scatter(ys[1], batch)) + ...: update_u of every output block in one launch
    (spherenet.py:219-225, the reference's accumulation order).  Linear in ys; its gradient is the same gathered row
    for every block (differentiable once: on the energy_and_force route the incoming gradient is the constant ones)."""
    return _GroupedGraphSum.apply(seg_batch, *ys)


def grouped_readout_supported(hidden, out_emb, out_channels, G):
    return (not _twice_differentiable and 1 <= G <= 8 and hidden in (32, 64, 128, 256) and out_emb % 8 == 0
            and 1 <= out_channels <= 8)


def grouped_readout(pairs, blocks, g):
    """u [B, out] from the (lin_rbf(rbf), e1) factor pair of every layer (their product is the reference's e2) and the
    matching output blocks (``lin_up``, ``lins``, ``lin``; swish)."""
    G = len(pairs)
    vs = _GroupedSegSum.apply(g.seg_dst, G, *[p[0] for p in pairs], *[p[1] for p in pairs])
    layers = [[(b.lin_up.weight, b.lin_up.bias, ACT_NONE, 0)] + [(lin.weight, lin.bias, ACT_SWISH, 0) for lin in b.lins]
              for b in blocks]
    if _wide_chain and wide_chain_supported(list(vs), layers):
        # lin_up and the three lins of ALL blocks: one launch per pass on row tiles that stay on the CU (csrc/wide.hip)
        hs = wide_chain(list(vs), layers)
    else:
        hs = _GroupedLinear.apply(ACT_NONE, G, *vs, *[b.lin_up.weight for b in blocks], *[b.lin_up.bias for b in blocks])
        for j in range(len(blocks[0].lins)):
            hs = _GroupedLinear.apply(ACT_SWISH, G, *hs, *[b.lins[j].weight for b in blocks],
                                      *[b.lins[j].bias for b in blocks])
    ys = _GroupedSmallN.apply(G, *hs, *[b.lin.weight for b in blocks], *[b.lin.bias for b in blocks])
    return _GroupedGraphSum.apply(g.seg_batch, *ys)


# ---------------------------------------------------------------------------------------------------
# fused triplet interaction (csrc/triplet.hip): basis -> first Linear of every layer in one pass, second
# Linear + gather + products + segment sum in one pass per layer.  No [T, ns*nr] / [T, ns^2*nr] /
# [T, int_emb] tensor is ever written.
# ---------------------------------------------------------------------------------------------------
PB, PO = 8, 32          # csrc/triplet.hip


def _stack_pad_t(ws):
    """list of <=4 weights [bs<=8, K] -> [K, 32] with column l*8+b = ws[l][b, :] (zeros elsewhere)."""
    K = ws[0].size(1)
    rows = []
    for w in ws:
        rows.append(w if w.size(0) == PB else torch.nn.functional.pad(w, (0, 0, 0, PB - w.size(0))))
    if len(ws) * PB < PO:
        rows.append(ws[0].new_zeros(PO - len(ws) * PB, K))
    return torch.cat(rows, 0).t().contiguous()


# route of the first basis Linears (dig3d_basis_project / dig3d_basis_wgrad): False = matrix cores where covered, True = the
# VALU kernels everywhere (tests and bench.py --route basis_valu=1 compare the two on one box)
basis_valu = False


class _BasisProject(Function):
    """(Ps_0..Ps_{L-1}[, Pt_0..Pt_{L-1}]) = first basis Linears of L <= 4 layers applied to the basis rows,
    which are evaluated on the fly (spherenet/features.py:213-222,256-263 + spherenet.py:163,166)."""

    @staticmethod
    def forward(ctx, bes, angle, torsion, kj, pref, cnt, ns, nr, nl, *weights):
        T = angle.numel()
        tor = torsion is not None
        dev = angle.device
        ws_l = [_f32c(w) for w in weights[:nl]]
        wt_l = [_f32c(w) for w in weights[nl:2 * nl]] if tor else None
        KS_, KT_ = ws_l[0].size(1), (wt_l[0].size(1) if tor else 0)
        Ws = torch.empty(KS_, PO, dtype=torch.float32, device=dev)
        Wt = torch.empty(KT_, PO, dtype=torch.float32, device=dev) if tor else None
        PP, IA = ctypes.c_void_p * nl, ctypes.c_int * nl
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        call('dig3d_basis_stack', nl, cast(PP(*[ptr(w) for w in ws_l])), cast(PP(*[ptr(w) for w in wt_l])) if tor else None,
             cast(IA(*[w.size(0) for w in ws_l])), cast(IA(*[w.size(0) for w in wt_l])) if tor else None, KS_, KT_,
             ptr(Ws), ptr(Wt), _stream())
        Ps = torch.empty(nl, T, PB, dtype=torch.float32, device=dev)
        Pt = torch.empty(nl, T, PB, dtype=torch.float32, device=dev) if tor else None
        call('dig3d_basis_project', ptr(bes), ptr(kj), ptr(angle), ptr(torsion), T, ns, nr, ptr(pref), ptr(Ws),
             ptr(Wt), nl, ptr(Ps), ptr(Pt), ptr(cnt), int(basis_valu), _stream())
        ctx.save_for_backward(bes, angle, torsion, kj, pref, cnt)
        ctx.leaf = _all_leaf(weights)
        # where the consumers of P_l (the fused triplet interaction of layer l) may write their gradient directly: slices
        # of ONE [nl, T, 8] buffer per table, so the backward below needs no torch.stack (2 framework launches per step)
        ctx.holder = dict(s=None, t=None, nl=nl, T=T)
        ctx.meta = (ns, nr, nl, [w.size(0) for w in weights[:nl]], [w.size(0) for w in weights[nl:2 * nl]])
        outs = tuple(Ps.unbind(0)) + (tuple(Pt.unbind(0)) if tor else ())
        for i, o in enumerate(outs):
            o._dig3d_gslot = (ctx.holder, 's' if i < nl else 't', i % nl)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        bes, angle, torsion, kj, pref, cnt = ctx.saved_tensors
        ns, nr, nl, bs_s, bs_t = ctx.meta
        tor = torsion is not None
        T = angle.numel()
        dev = angle.device

        def stack(gs):
            z = None
            out = []
            for gr in gs:
                if gr is None:
                    if z is None:
                        z = torch.zeros(T, PB, dtype=torch.float32, device=dev)
                    gr = z
                out.append(gr)
            return torch.stack(out, 0).contiguous()

        def from_slots(gs, key):
            buf = ctx.holder[key]
            if buf is None or any(g is None or g.data_ptr() != buf[l].data_ptr() or g.shape != buf[l].shape
                                  for l, g in enumerate(gs)):
                return stack(gs)
            return buf

        gPs = from_slots(grads[:nl], 's')
        gPt = from_slots(grads[nl:2 * nl], 't') if tor else None
        KS, KT = ns * nr, (ns * ns * nr if tor else 0)
        nb = _hip.query('dig3d_basis_wgrad_blocks', T)
        part = torch.empty(nb * (KS + KT) * PO, dtype=torch.float32, device=dev)
        gWs = torch.empty(PO, KS, dtype=torch.float32, device=dev)       # row l*8 + b = weight row b of layer l
        gWt = torch.empty(PO, KT, dtype=torch.float32, device=dev) if tor else None
        n = (KS + KT) * PO
        # (T == 0: the kernel does not run and writes no partial — the entry point zero-fills the gradients itself; a
        # deferred reduction would sum the unwritten scratch into them.  Found by the poisoned-allocation sweep of r04.)
        if _deferred is not None and ctx.leaf and T > 0:       # reduced with every other layer's partials in one launch
            now = 0
            _deferred.add(part, nb, n, gWs, KS * PO)
            if tor:
                _deferred.add(part[KS * PO:], nb, n, gWt, KT * PO)
        else:
            now = 1
        call('dig3d_basis_wgrad', ptr(bes), ptr(kj), ptr(angle), ptr(torsion), T, ns, nr, ptr(pref), ptr(gPs),
             ptr(gPt), nl, ptr(part), ptr(gWs), ptr(gWt), ptr(cnt), now, int(basis_valu), _stream())
        gw = [gWs[l * PB:l * PB + bs_s[l]] for l in range(nl)]
        if tor:
            gw += [gWt[l * PB:l * PB + bs_t[l]] for l in range(nl)]
        return (None,) * 9 + tuple(gw)


def basis_project(bes, angle, torsion, kj, pref, ns, nr, w_sbf1, w_t1=None, cnt=None):
    """-> (Ps, Pt): lists (one [T,8] tensor per layer) of lin_sbf1 / lin_t1 applied to the on-the-fly basis.
    Layers are processed in groups of 4 (32 stacked outputs per launch)."""
    Ps, Pt = [], []
    L = len(w_sbf1)
    for a in range(0, L, PO // PB):
        ws = list(w_sbf1[a:a + PO // PB])
        wt = list(w_t1[a:a + PO // PB]) if w_t1 is not None else []
        outs = _BasisProject.apply(bes, angle, torsion, kj, pref, cnt, ns, nr, len(ws), *ws, *wt)
        Ps += list(outs[:len(ws)])
        Pt += list(outs[len(ws):])
    return Ps, (Pt if w_t1 is not None else None)


def _pad8(w):
    w = _f32c(w)
    return w if w.size(1) == PB else torch.nn.functional.pad(w, (0, PB - w.size(1))).contiguous()


# route of the fused triplet interaction (dig3d_triplet_fwd / dig3d_triplet_bwd, include/dig3d.h): 0 = a wave per segment where
# covered (C = 64 / 128 / 256), in the form measured best for the size; 1 = the lane-group kernels everywhere; 2 / 3 = the
# scalar-operand / index-chain-once forms of the wave kernels always (tests and bench.py --route trip_lane_groups=N compare)
trip_lane_groups = 0
# energy_and_force route of a model WITHOUT torsion (DimeNet++): True = the fused triplet kernels as a family closed under
# differentiation (dig_amd/diffops.py:trip2), False = the round-2 route (basis table x composed Linear [T, 42] -> [T, int_emb],
# then gather-multiply-segment-sum); bench.py --route force_trip2=0 compares on one box
force_trip2 = True
# ... and the front of every block (lin_ji, lin_kj, the product with the radial projection, lin_down) as ONE twice-differentiable
# launch per pass (dig_amd/diffops.py:front2, csrc/chain.hip:k_front_dd); False = grouped pair + mul2 + lin_down Functions
force_front2 = True
# ... and the output blocks (lin_up + lins of all L + 1 blocks) as ONE twice-differentiable launch per pass on the 256-wide chain
# kernels (dig_amd/diffops.py:wide2); False = one grouped twice-differentiable launch per stage
force_wide2 = True
# ... and the angular basis contracted with lin_sbf1 of all blocks inside the basis kernels (dig_amd/diffops.py:sbf_project,
# csrc/sbf2.hip); False = the [T, ns nr] table by framework broadcasting + a stacked T-row dense layer
force_sbf_fused = True
# ComENet blocks below this many nodes run their pairs of independent layers as grouped launches (launch-latency regime);
# above it the per-layer persistent kernels are the better ones (config 5: 16 384 rows)
comenet_group_rows = 4096
comenet_group_pairs = True        # ... (False: per-layer launches there too; bench.py --route comenet_group_pairs=0 compares)
comenet_wide_single = True        # ... and their 256-wide layers go through the 256-wide chain kernel (csrc/wide.hip)
# dig3d_wgrad_many route 1: two staging buffers, one barrier per chunk — bit-identical partials; same-box A/B (r05): config 2
# 1.5160 -> 1.5105 ms, config 4 5.399 -> 5.378, config 5 6.950 -> 6.928 (bench.py --route wgrad_double_buffer=0 compares)
wgrad_double_buffer = True
wgrad_blocks_per_cu = 2           # blocks per CU one deferred weight-gradient launch is sized for (two fit the LDS)
edge_front_fused = True           # edge lengths + dist_emb + Bessel table of the energy route as one launch (diffops.edge_front)
schnet_group_filters = True       # SchNet: the first filter-generating layer of all blocks as one grouped launch per pass
force_group_segsum = True         # the edge -> node sums of all output blocks as one launch per pass (diffops.segsum_grouped)
force_mul_segsum = True           # e2 = lin_rbf(rbf) * e1 formed inside the grouped edge -> node sums (diffops.mul_segsum_grouped)
force_radial2 = True              # the blocks' 2 L radial projections as one closed family on csrc/radial.hip (diffops.radial2)
force_group_radial = True         # the blocks' 2 L radial projections as one grouped twice-differentiable launch per pass
force_group_front = True          # lin_ji + lin_kj (same input) as one grouped twice-differentiable launch per pass
force_trip2_stacked = True        # lin_sbf1 of all blocks as one stacked T-row layer (False: one layer per block)
force_fan_out = True              # rbf's 2 + 2 L consumers get aliases; their gradients are summed by one launch per pass


class _TripletInteraction(Function):
    """out[e] = sum_{t: ji[t]=e} X[kj[t]] * (W2s Ps[t]) * (W2t Pt[t])  (spherenet.py:164-171, dimenetpp.py:147-150)."""

    @staticmethod
    def _slot(sl, T, dev):
        """the [T, 8] gradient of one projected basis: a row of the layer-stacked buffer the basis weight-gradient
        kernel reads when the forward handed one out (``_dig3d_gslot``), else a buffer of its own"""
        if sl is None:
            return torch.empty(T, PB, dtype=torch.float32, device=dev)
        holder, key, l = sl
        if holder['T'] != T:
            return torch.empty(T, PB, dtype=torch.float32, device=dev)
        if holder[key] is None:
            holder[key] = torch.empty(holder['nl'], T, PB, dtype=torch.float32, device=dev)
        return holder[key][l]

    @staticmethod
    def forward(ctx, X, Ps, Pt, W2s, W2t, g):
        X, Ps = _f32c(X), _f32c(Ps)
        tor = Pt is not None
        Pt = _f32c(Pt) if tor else None
        w2s, w2t = _pad8(W2s), (_pad8(W2t) if tor else None)
        E, C = X.shape
        if Ps.size(0) == 0 or E == 0:            # no triplets at all: every segment is empty
            out = torch.zeros(E, C, dtype=torch.float32, device=X.device)
        else:
            out = torch.empty(E, C, dtype=torch.float32, device=X.device)
            call('dig3d_triplet_fwd', ptr(X), ptr(g.kj), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(g.tptr), None, E, C,
                 ptr(out), int(trip_lane_groups), _stream())
        ctx.g, ctx.bs = g, (W2s.size(1), W2t.size(1) if tor else 0)
        ctx.leaf = _all_leaf((W2s, W2t))
        ctx.slots = (getattr(Ps, '_dig3d_gslot', None), getattr(Pt, '_dig3d_gslot', None) if tor else None)
        ctx.save_for_backward(X, Ps, Pt, w2s, w2t)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, G):
        X, Ps, Pt, w2s, w2t = ctx.saved_tensors
        g = ctx.g
        tor = Pt is not None
        G = _f32c(G)
        E, C = X.shape
        T = Ps.size(0)
        dev = X.device
        if T == 0 or E == 0:
            z8 = torch.zeros(T, PB, dtype=torch.float32, device=dev)
            zw = torch.zeros(C, PB, dtype=torch.float32, device=dev)
            return (torch.zeros_like(X), z8, (z8 if tor else None), zw[:, :ctx.bs[0]],
                    (zw[:, :ctx.bs[1]] if tor else None), None)
        gX = None
        if ctx.needs_input_grad[0]:
            seg = g.seg_kj
            gX = torch.empty_like(X)
            call('dig3d_triplet_fwd', ptr(G), ptr(g.ji), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(seg.kptr),
                 ptr(seg.perm), E, C, ptr(gX), int(trip_lane_groups), _stream())
        gPs = _TripletInteraction._slot(ctx.slots[0], T, dev)
        gPt = _TripletInteraction._slot(ctx.slots[1], T, dev) if tor else None
        nb = _hip.query('dig3d_triplet_bwd_blocks', E, C, int(trip_lane_groups))
        part = torch.empty(nb * 2 * C * PB, dtype=torch.float32, device=dev)
        gW2s = torch.empty(C, PB, dtype=torch.float32, device=dev)
        gW2t = torch.empty(C, PB, dtype=torch.float32, device=dev) if tor else None
        if _deferred is not None and ctx.leaf:
            now = 0
            _deferred.add(part, nb, 2 * C * PB, gW2s, C * PB)
            if tor:
                _deferred.add(part[C * PB:], nb, 2 * C * PB, gW2t, C * PB)
        else:
            now = 1
        call('dig3d_triplet_bwd', ptr(G), ptr(X), ptr(g.kj), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(g.tptr), E, C,
             ptr(gPs), ptr(gPt), ptr(part), ptr(gW2s), ptr(gW2t), now, int(trip_lane_groups), _stream())
        bs_s, bs_t = ctx.bs
        return gX, gPs, gPt, gW2s[:, :bs_s], (gW2t[:, :bs_t] if tor else None), None


def triplet_interaction(X, Ps, Pt, W2s, W2t, g):
    return _TripletInteraction.apply(X, Ps, Pt, W2s, W2t, g)


def triplet_fused_supported(C, ns, nr, basis_sizes, torsion):
    """Shapes the fused kernels cover (everything else takes the table + GEMM route)."""
    K = ns * nr + (ns * ns * nr if torsion else 0)
    return C in (16, 32, 64, 128, 256) and K <= 384 and max(basis_sizes) <= PB and 1 <= ns <= 8


# the seed of the backward pass when the caller knows it before the loss is formed (dig_amd/graphed.py: a device scalar the
# captured step reads): _L1Mean then writes its gradient in the forward launch
_loss_seed = None


class known_loss_seed:
    """``with known_loss_seed(seed): loss = loss_fn(...)`` — ``seed`` is the 0-dim float32 tensor that will be passed as
    ``grad_outputs`` of this loss."""

    def __init__(self, seed):
        self.seed = seed

    def __enter__(self):
        global _loss_seed
        self.prev, _loss_seed = _loss_seed, self.seed

    def __exit__(self, *a):
        global _loss_seed
        _loss_seed = self.prev


class _L1Mean(Function):
    """mean |out - target| (torch.nn.L1Loss(), run.py:49,127) and its gradient in two launches (csrc/readout.hip) — one when
    the backward seed is known at forward time (``known_loss_seed``)."""

    @staticmethod
    def forward(ctx, out, target):
        out = _f32c(out)
        target = _f32c(target.expand_as(out))
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        sgn = torch.empty_like(out)
        seed = _loss_seed
        ctx.pre = None
        if seed is not None and seed.is_cuda and seed.dtype == torch.float32 and seed.numel() == 1:
            ctx.pre = (seed.data_ptr(), torch.empty_like(out), seed, seed._version)
        call('dig3d_l1_loss_fwd', ptr(out), ptr(target), out.numel(), ptr(loss), ptr(sgn),
             ptr(seed) if ctx.pre else None, ptr(ctx.pre[1]) if ctx.pre else None, _stream())
        ctx.save_for_backward(sgn)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        (sgn,) = ctx.saved_tensors
        if (ctx.pre is not None and gl.is_cuda and gl.numel() == 1 and gl.data_ptr() == ctx.pre[0]
                and ctx.pre[2]._version == ctx.pre[3]):
            # the incoming gradient IS the announced seed, unmodified since (an in-place write to the seed between
            # forward and backward bumps its version and takes the general path): written by the forward launch
            return ctx.pre[1], None
        g = torch.empty_like(sgn)
        call('dig3d_scale_by_scalar', ptr(sgn), ptr(_f32c(gl)), sgn.numel(), ptr(g), _stream())
        return g, None


class _EFL1Loss(Function):
    """mean |out - y| + p * mean |(-gpos) - f| (run.py:126-131 with torch.nn.L1Loss(); force = -gpos) in ONE launch
    (csrc/readout.hip:k_ef_l1_loss) — and, when the backward seed is known at forward time, both gradients with it.
    ``gpos`` carries the create_graph graph of the position gradient; this Function is differentiated once."""

    @staticmethod
    def forward(ctx, out, y, gpos, f, cnt_n, p):
        out, gpos = _f32c(out), _f32c(gpos)
        y, f = _f32c(y.expand_as(out)), _f32c(f)
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        sgn_e, sgn_f = torch.empty_like(out), torch.empty_like(gpos)
        seed = _loss_seed
        ctx.pre = None
        if seed is not None and seed.is_cuda and seed.dtype == torch.float32 and seed.numel() == 1:
            ctx.pre = (seed.data_ptr(), seed._version, torch.empty_like(out), torch.empty_like(gpos))
        call('dig3d_ef_l1_loss', ptr(out), ptr(y), out.numel(), ptr(gpos), ptr(f), gpos.numel(), ptr(cnt_n), float(p),
             ptr(seed) if ctx.pre else None, ptr(loss), ptr(sgn_e), ptr(sgn_f), ptr(ctx.pre[2]) if ctx.pre else None,
             ptr(ctx.pre[3]) if ctx.pre else None, _stream())
        ctx.seed = seed if ctx.pre else None
        ctx.save_for_backward(sgn_e, sgn_f)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        sgn_e, sgn_f = ctx.saved_tensors
        if (ctx.pre is not None and gl.is_cuda and gl.numel() == 1 and gl.data_ptr() == ctx.pre[0]
                and ctx.seed._version == ctx.pre[1]):
            return ctx.pre[2], None, ctx.pre[3], None, None, None      # the incoming gradient IS the announced seed
        gl = _f32c(gl)
        ge, gf = torch.empty_like(sgn_e), torch.empty_like(sgn_f)
        call('dig3d_scale_by_scalar', ptr(sgn_e), ptr(gl), sgn_e.numel(), ptr(ge), _stream())
        call('dig3d_scale_by_scalar', ptr(sgn_f), ptr(gl), sgn_f.numel(), ptr(gf), _stream())
        return ge, None, gf, None, None, None


def ef_l1_loss(out, y, gpos, force_target, cnt_n=None, p=100.0):
    """``l1(out, y) + p * l1(-gpos, force_target)`` with mean reduction over the live atoms (``cnt_n``: device int32 live node
    count of a padded batch, None: all rows) — the trainer's energy_and_force loss as one kernel."""
    return _EFL1Loss.apply(out, y, gpos, force_target, cnt_n, p)


def l1_mean(out, target):
    """torch.nn.functional.l1_loss(out, target) for float32 GPU tensors of <= 2^20 entries with a constant target; the
    framework op otherwise (CPU tensors included: the loss of a user-supplied pipeline is not the engine's business)."""
    if (out.is_cuda and out.dtype == torch.float32 and target.dtype == torch.float32 and not target.requires_grad
            and 0 < out.numel() <= (1 << 20) and not _twice_differentiable and target.numel() in (1, out.numel())):
        return _L1Mean.apply(out, target.reshape(out.shape) if target.numel() == out.numel() else target)
    return torch.nn.functional.l1_loss(out, target.expand_as(out) if target.numel() == 1 else target)


class _GraphNorm(Function):
    """PyG GraphNorm (comenet.py:160,213) as one kernel forward, one backward (csrc/norm.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean_scale, gptr, B, eps, padded=False):
        x = _f32c(x)
        N, C = x.shape
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(B, C, dtype=torch.float32, device=dev)
        rstd = torch.empty(B, C, dtype=torch.float32, device=dev)
        call('dig3d_graphnorm_fwd', ptr(x), ptr(gptr), B, C, ptr(weight), ptr(bias), ptr(mean_scale), float(eps), ptr(y),
             ptr(mean), ptr(rstd), _stream())
        if padded:          # static-shape batch: the node rows behind the last graph are written by nobody else
            call('dig3d_zero_rows_from', ptr(y), gptr.data_ptr() + 4 * B, N, C, _stream())
        ctx.B, ctx.padded = B, padded
        ctx.leaf = _all_leaf((weight, bias, mean_scale))
        ctx.save_for_backward(x, weight, mean_scale, mean, rstd, gptr)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight, mean_scale, mean, rstd, gptr = ctx.saved_tensors
        gy = _f32c(gy)
        N, C = x.shape
        B = ctx.B
        gx = torch.empty_like(x)
        part = torch.empty(max(B, 1) * 3 * C, dtype=torch.float32, device=x.device)
        gp = torch.empty(3 * C, dtype=torch.float32, device=x.device)
        # the per-graph partials of the three parameter gradients join the deferred reductions of the pass when there is one
        now = _reduce_later(part, B, 3 * C, gp, ctx.leaf) if B > 0 else 1
        call('dig3d_graphnorm_bwd', ptr(gy), ptr(x), ptr(gptr), B, C, ptr(weight), ptr(mean_scale), ptr(mean), ptr(rstd),
             ptr(gx), ptr(part), ptr(gp) if now else None, _stream())
        if ctx.padded:
            call('dig3d_zero_rows_from', ptr(gx), gptr.data_ptr() + 4 * B, N, C, _stream())
        return gx, gp[:C], gp[C:2 * C], gp[2 * C:], None, None, None, None


class _ComposeWeights(Function):
    """Wc_p = W2_p W1_p for every bias-free two-layer projection of a model in one launch; one more for the gradients of
    all factors (csrc/dense.hip:dig3d_compose_fwd / _bwd; comenet.py:87-105)."""

    @staticmethod
    def forward(ctx, n, *ws):
        W2s = [_f32c(w) for w in ws[:n]]
        W1s = [_f32c(w) for w in ws[n:]]
        outs = [torch.empty(a.size(0), b.size(1), dtype=torch.float32, device=a.device) for a, b in zip(W2s, W1s)]
        IA = ctypes.c_int * n
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        p2, k2 = _ptrs(W2s)
        p1, k1 = _ptrs(W1s)
        po, ko = _ptrs(outs)
        ctx.dims = (IA(*[a.size(0) for a in W2s]), IA(*[a.size(1) for a in W2s]), IA(*[b.size(1) for b in W1s]))
        call('dig3d_compose_fwd', n, p2, p1, cast(ctx.dims[0]), cast(ctx.dims[1]), cast(ctx.dims[2]), po, _stream())
        ctx.n = n
        ctx.save_for_backward(*W2s, *W1s)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        n = ctx.n
        sv = ctx.saved_tensors
        W2s, W1s = sv[:n], sv[n:]
        gs = [_f32c(g) if g is not None else torch.zeros(a.size(0), b.size(1), dtype=torch.float32, device=a.device)
              for g, a, b in zip(gs, W2s, W1s)]
        g2 = [torch.empty_like(a) for a in W2s]
        g1 = [torch.empty_like(b) for b in W1s]
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        pg, kg = _ptrs(gs)
        p2, k2 = _ptrs(W2s)
        p1, k1 = _ptrs(W1s)
        q2, j2 = _ptrs(g2)
        q1, j1 = _ptrs(g1)
        call('dig3d_compose_bwd', n, pg, p2, p1, cast(ctx.dims[0]), cast(ctx.dims[1]), cast(ctx.dims[2]), q2, q1, _stream())
        return (None,) + tuple(g2) + tuple(g1)


def compose_weights(pairs):
    """[(W2 [N, Mid], W1 [Mid, K]), ...] (at most 16) -> [W2 W1, ...]: the weights of ``x -> lin2(lin1(x))`` for bias-free,
    activation-free layer pairs, all in one launch."""
    n = len(pairs)
    return list(_ComposeWeights.apply(n, *[p[0] for p in pairs], *[p[1] for p in pairs]))


def graph_norm(x, weight, bias, mean_scale, gptr, B, eps=1e-5, padded=False):
    """``padded``: x has rows behind the last graph (a static-shape batch) — they come out as zeros, forward and backward"""
    return _GraphNorm.apply(x, weight, bias, mean_scale, gptr, B, eps, padded)


# ---------------------------------------------------------------------------------------------------
# geometry + basis (forward only: constants w.r.t. the parameters)
# ---------------------------------------------------------------------------------------------------
def edge_dist(pos, g, mode=0):
    out = torch.empty(g.E, dtype=torch.float32, device=pos.device)
    # padded edges of a static-shape batch get dist = 1 (inside every cutoff: all basis values finite)
    call('dig3d_edge_dist', ptr(pos), ptr(g.src), ptr(g.dst), g.E, mode, ptr(out), ptr(g.cnt_E), 1.0, _stream())
    return out


def triplet_geom(pos, g, use_torsion):
    dev = pos.device
    angle = torch.empty(g.T, dtype=torch.float32, device=dev)
    torsion = torch.empty(g.T, dtype=torch.float32, device=dev) if use_torsion else None
    targ = torch.empty(g.T, dtype=torch.int32, device=dev) if use_torsion else None
    call('dig3d_triplet_geom', ptr(pos), ptr(g.rowptr), ptr(g.col), ptr(g.src), ptr(g.dst), ptr(g.kj), ptr(g.ji),
         g.T, int(bool(use_torsion)), ptr(angle), ptr(torsion), ptr(targ), ptr(g.cnt_T), _stream())
    return angle, torsion, targ


def bessel_basis(dist, cutoff, ns, nr, zeros_d, norms_d, envelope_p=0):
    E = dist.numel()
    out = torch.empty(E, ns * nr, dtype=torch.float32, device=dist.device)
    call('dig3d_bessel_basis', ptr(dist), E, float(cutoff), ns, nr, ptr(zeros_d), ptr(norms_d), int(envelope_p),
         ptr(out), _stream())
    return out


def sph_basis(bes, gidx, theta, phi, ns, nr, pref_d, pair_mode):
    M = theta.numel()
    H = ns if phi is None else ns * ns
    out = torch.empty(M, H * nr, dtype=torch.float32, device=theta.device)
    call('dig3d_sph_basis', ptr(bes), ptr(gidx), ptr(theta), ptr(phi), M, ns, nr, ptr(pref_d), int(pair_mode),
         ptr(out), _stream())
    return out


def segment_argmin(val, add, seg, sentinel):
    out_arg = torch.empty(seg.S, dtype=torch.int32, device=val.device)
    out_val = torch.empty(seg.S, dtype=torch.float32, device=val.device)
    call('dig3d_segment_argmin', ptr(val), ptr(add), ptr(seg.kptr), ptr(seg.perm), seg.S, int(sentinel),
         ptr(out_val), ptr(out_arg), _stream())
    return out_val, out_arg


# ---------------------------------------------------------------------------------------------------
# public, reference-shaped API
# ---------------------------------------------------------------------------------------------------
def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow='source_to_target',
                 num_workers=1):
    """torch_cluster.radius_graph drop-in (spherenet.py:304).  Returns int64 [2, E], row 0 = source,
    row 1 = target, grouped by target ascending, sources ascending (the CUDA ordering rule)."""
    if flow != 'source_to_target':
        raise NotImplementedError("only flow='source_to_target' (the only one DIG uses)")
    return build_graph(x, batch, r, max_num_neighbors, loop, triplets=False).edge_index


def _seg_from_index(index, S):
    """Seg for an int64 index: sorted -> CSR by scan of counts, unsorted -> transposed CSR."""
    M = index.numel()
    key = torch.empty(max(M, 1), dtype=torch.int32, device=index.device)[:M]
    if M:
        call('dig3d_cast_i64_i32', ptr(index.contiguous()), ptr(key), M, _stream())
    return csr_by_key(key, S)


class _ScatterMeanSorted(Function):
    """scatter(src, sorted int64 index, reduce='mean'): one pass, every byte read once."""

    @staticmethod
    def forward(ctx, src, index, S):
        src = _f32c(src)
        M, C = src.shape
        out = torch.empty(S, C, dtype=torch.float32, device=src.device)
        call('dig3d_segment_mean_sorted', ptr(src), ptr(index), M, C, S, ptr(out), _stream())
        ctx.save_for_backward(index)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        seg = _seg_from_index(index, ctx.S)
        g = _f32c(g)
        gs = torch.empty_like(g)
        call('dig3d_rows_div_count', ptr(g), ptr(seg.kptr), seg.S, g.size(1), ptr(gs), _stream())
        return _Gather.apply(gs, seg), None, None


class _ScatterSumSorted(Function):
    @staticmethod
    def forward(ctx, src, index, S, tuning):
        src = _f32c(src)
        M, C = src.shape
        out = torch.empty(S, C, dtype=torch.float32, device=src.device)
        if tuning is None:
            call('dig3d_segment_sum_sorted', ptr(src), ptr(index), M, C, S, ptr(out), _stream())
        else:           # (rows per worker, kernel variant): sweeps and tiling-independence tests
            call('dig3d_segment_sum_sorted_tuned', ptr(src), ptr(index), M, C, S, ptr(out), int(tuning[0]),
                 int(tuning[1]), _stream())
        ctx.save_for_backward(index)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return _Gather.apply(g, _seg_from_index(index, ctx.S)), None, None, None


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum', assume_sorted=None, tuning=None):
    """torch_scatter.scatter drop-in for the forms DIG uses: 1-D index along ``dim`` of a 1-D or 2-D
    ``src`` (dim = 0 for 2-D), reduce in {'sum','add','mean','min'}.  ``dim_size=None`` costs a host sync
    (index.max()), exactly like the original."""
    if out is not None:
        raise NotImplementedError('out= is never used by dig.threedgraph')
    if index.dim() != 1:
        raise NotImplementedError('only 1-D index (the reference never broadcasts a multi-dim index)')
    d = dim if dim >= 0 else src.dim() + dim
    if not ((src.dim() == 1 and d == 0) or (src.dim() == 2 and d == 0)):
        raise NotImplementedError('scatter along dim 0 of a 1-D/2-D tensor only')
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    if reduce == 'min':
        return scatter_min(src, index, dim, None, dim_size)[0]
    if reduce not in ('sum', 'add', 'mean'):
        raise ValueError(reduce)
    if index.numel() != src.size(0):
        raise RuntimeError(f'scatter: index has {index.numel()} entries but src has {src.size(0)} rows')
    x = src.unsqueeze(1) if src.dim() == 1 else src
    if assume_sorted is None:
        assume_sorted = bool((index[1:] >= index[:-1]).all()) if index.numel() > 1 else True
    if reduce == 'mean':        # division by the segment length fused into the kernel's store (csrc/segment.hip)
        if assume_sorted:
            res = _ScatterMeanSorted.apply(x, index.contiguous(), dim_size)
        else:
            res = _SegMean.apply(x, _seg_from_index(index, dim_size))
    elif assume_sorted:
        res = _ScatterSumSorted.apply(x, index.contiguous(), dim_size, tuning)
    else:
        res = _SegSum.apply(x, _seg_from_index(index, dim_size))
    return res.squeeze(1) if src.dim() == 1 else res


class _ScatterMin(Function):
    @staticmethod
    def forward(ctx, src, index, S):
        seg = _seg_from_index(index, S)
        E = src.numel()
        val, arg = segment_argmin(_f32c(src), None, seg, E)
        arg64 = arg.to(torch.int64)
        ctx.save_for_backward(arg64)
        ctx.E = E
        ctx.mark_non_differentiable(arg64)
        return val, arg64

    @staticmethod
    def backward(ctx, g, _):
        (arg,) = ctx.saved_tensors
        # each segment's minimum came from exactly one entry: a scatter with unique targets (sentinel arg == E: none)
        g = _f32c(g)
        gs = torch.empty(ctx.E, dtype=torch.float32, device=g.device)
        call('dig3d_scatter_unique', ptr(g), ptr(arg), arg.numel(), ctx.E, ptr(gs), _stream())
        return gs, None, None


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    """torch_scatter.scatter_min for 1-D src: (min per segment, FIRST arg-min; empty -> (0, len(src)))."""
    if src.dim() != 1 or index.dim() != 1 or out is not None:
        raise NotImplementedError('1-D scatter_min only (comenet.py:304-325, geometric_computing.py:75)')
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    return _ScatterMin.apply(src, index, dim_size)
