"""Per-batch molecular graph resident in HBM, built by the HIP engine (csrc/graph.hip).

One ``MolGraph`` replaces what the reference recomputes with PyG / torch_sparse calls on every forward:
``radius_graph`` (spherenet.py:304), the ``SparseTensor`` CSR and the triplet index lists of
``xyz_to_dat`` (utils/geometric_computing.py:27-41).  Layout: edges grouped by target node (ascending),
sources ascending inside a target — i.e. the edge list is its own CSR, and every forward reduction of the
models (edge->node by ``dst``, triplet->edge by ``ji``, node->graph by ``batch``) is a contiguous segment
sum.  Transposed CSRs (for the backward of the gathers by ``src`` and ``kj``) are built lazily, once.

Exactly ONE device->host copy (B, E, T) happens per batch.
"""
import ctypes

import torch

from . import _hip
from ._hip import call, ptr


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Seg:
    """A segmentation of M rows into S segments: ``key[M]`` (segment id per row, int32) and its CSR
    ``kptr[S+1]``.  ``perm`` is None when ``key`` is sorted (rows of a segment are contiguous), else the
    row order grouped by key (transposed CSR, ascending inside a key)."""
    __slots__ = ('key', 'kptr', 'perm', 'S', 'M', 'cnt', 'aux')

    def __init__(self, key, kptr, perm, S, cnt=None):
        self.key, self.kptr, self.perm, self.S, self.M = key, kptr, perm, int(S), int(key.numel())
        # cnt: device int32 scalar = live row count when ``key`` is padded to a static capacity (HIP-graph
        # batches, dig_amd/graphed.py); None for exact-size batches.
        self.cnt = cnt
        self.aux = None     # per-graph scratch of the ops (e.g. operands permuted into this grouping's order, built once)


def csr_by_key(key, S):
    """Seg for an arbitrary (unsorted) int32 key in [0, S)."""
    dev = key.device
    M = key.numel()
    kptr = torch.empty(S + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    hc = torch.empty(2 * max(S, 1), dtype=torch.int32, device=dev)      # histogram + cursors: adjacent, zeroed by one memset
    hist, cursor = hc[:max(S, 1)], hc[max(S, 1):]
    tmp = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    ws = torch.empty(S // 4096 + 3, dtype=torch.int32, device=dev)
    if S == 0:
        kptr.zero_()
    call('dig3d_csr_by_key', ptr(key), M, S, ptr(kptr), ptr(perm), ptr(hist), ptr(cursor), ptr(tmp), ptr(ws), _stream())
    return Seg(key, kptr, perm[:M], S)


_hc_ws = {}     # (device, stream) -> int32 workspace that dig3d_csr_by_keys_ws leaves all zero
keep_hc_workspace = True   # False: a fresh workspace and a zero fill per call (tests / A-B)


def csr_by_keys(items):
    """[(key, S), ...] (at most 4, every S <= 32768) -> [Seg, ...]: the transposed CSRs of several keys in one set of
    launches (csrc/graph.hip:dig3d_csr_by_keys); falls back to one ``csr_by_key`` per key otherwise."""
    if not items or len(items) > 4 or any(S > 32768 or S < 1 for _, S in items):
        return [csr_by_key(k, S) for k, S in items]
    dev = items[0][0].device
    n = len(items)
    i32 = dict(dtype=torch.int32, device=dev)
    Ms = [k.numel() for k, _ in items]
    Ss = [S for _, S in items]
    kptrs = [torch.empty(S + 1, **i32) for S in Ss]
    perms = [torch.empty(max(M, 1), **i32) for M in Ms]
    tmps = [torch.empty(max(M, 1), **i32) for M in Ms]
    # histograms + cursors of all keys: ONE workspace per device and stream, kept between batches — the launch set leaves it
    # zero, so only its first use pays a zero fill (was a fill launch per batch)
    words = 2 * sum(Ss)
    wkey = (dev, _stream())
    hc = _hc_ws.get(wkey)
    clean = 1
    # (inside a HIP-graph capture an allocation belongs to the graph's private pool: never kept, never reused there)
    capturing = torch.cuda.is_current_stream_capturing()
    if hc is None or hc.numel() < words or not keep_hc_workspace or capturing:
        if keep_hc_workspace and not capturing:
            hc = torch.zeros(max(words, 1 << 17), **i32)     # the WHOLE workspace zero once: later calls use other extents
            _hc_ws[wkey] = hc
        else:
            hc = torch.empty(words, **i32)
            clean = 0
    offs = [2 * sum(Ss[:i]) for i in range(n)]
    PP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
    try:
        call('dig3d_csr_by_keys_ws', n, cast(PP(*[k.data_ptr() for k, _ in items])), cast(IA(*Ms)), cast(IA(*Ss)),
             cast(PP(*[t.data_ptr() for t in kptrs])), cast(PP(*[t.data_ptr() for t in perms])),
             cast(PP(*[hc.data_ptr() + 4 * o for o in offs])), cast(PP(*[t.data_ptr() for t in tmps])), clean, _stream())
    except Exception:
        _hc_ws.pop(wkey, None)            # state unknown: the next call starts from a fresh workspace
        raise
    return [Seg(k, kptr, perm[:M], S) for (k, S), kptr, perm, M in zip(items, kptrs, perms, Ms)]


class MolGraph:
    def __init__(self):
        self.N = self.B = self.E = self.T = 0
        self.composite = False
        self.cnt_N = self.cnt_E = self.cnt_T = None      # device live counts of a padded (static-shape) graph
        self._by_src = self._by_kj = self._by_dst = self._edge_index = self._idx64 = None

    # --- segmentations used by the models -------------------------------------------------------
    @property
    def seg_dst(self):          # edges -> target node (sorted for engine-built graphs)
        if getattr(self, '_sorted_edges', True):
            return Seg(self.dst, self.rowptr, None, self.N, self.cnt_E)
        if self._by_dst is None:
            self._by_dst = csr_by_key(self.dst, self.N)
        return self._by_dst

    @property
    def seg_src(self):          # edges -> source node (unsorted; transposed CSR built on first use)
        if self._by_src is None:
            self._by_src = csr_by_key(self.src, self.N)
        return self._by_src

    def build_transposed(self, triplets=True):
        """seg_src (and seg_kj) built together: one set of launches instead of one per key"""
        want = [('_by_src', self.src, self.N)] if self._by_src is None else []
        if triplets and self._by_kj is None and getattr(self, 'kj', None) is not None:
            want.append(('_by_kj', self.kj, self.E))
        want = [w for w in want if w[2] > 0]
        if want:
            for (name, _, _), seg in zip(want, csr_by_keys([(k, S) for _, k, S in want])):
                setattr(self, name, seg)

    @property
    def seg_ji(self):           # triplets -> edge j->i (sorted)
        return Seg(self.ji, self.tptr, None, self.E, self.cnt_T)

    @property
    def seg_kj(self):           # triplets -> edge k->j (unsorted)
        if self._by_kj is None:
            self._by_kj = csr_by_key(self.kj, self.E)
        return self._by_kj

    @property
    def seg_batch(self):        # nodes -> graph (sorted)
        return Seg(self.batch32, self.ptr, None, self.B, self.cnt_N)

    # --- int64 views for the public API -----------------------------------------------------------
    @property
    def edge_index(self):
        """int64 [2, E], row 0 = source j, row 1 = target i (torch_cluster.radius_graph layout)."""
        if self._edge_index is None:
            ei = torch.empty(2, self.E, dtype=torch.int64, device=self.src.device)
            if self.E:
                call('dig3d_cast_i32_i64', ptr(self.src), ptr(ei[0]), self.E, _stream())
                call('dig3d_cast_i32_i64', ptr(self.dst), ptr(ei[1]), self.E, _stream())
            self._edge_index = ei
        return self._edge_index

    @property
    def idx_kj_ji(self):
        if self._idx64 is None:
            dev = self.src.device
            kj = torch.empty(self.T, dtype=torch.int64, device=dev)
            ji = torch.empty(self.T, dtype=torch.int64, device=dev)
            if self.T:
                call('dig3d_cast_i32_i64', ptr(self.kj), ptr(kj), self.T, _stream())
                call('dig3d_cast_i32_i64', ptr(self.ji), ptr(ji), self.T, _stream())
            self._idx64 = (kj, ji)
        return self._idx64


_pinned_meta = []          # pool of pinned int64[8] host buffers for the asynchronous (B, E, T) read-back


class PendingGraph:
    """A graph build whose size-dependent second stage has not run yet: stage 1 (radius search, CSR, triplet
    counts) is enqueued and its (B, E, T) is on its way to pinned host memory; ``finish()`` waits for exactly that
    copy and completes the build.  Lets the build of batch i+1 be queued behind the replay of batch i, so the host
    work of the prologue overlaps GPU execution (dig_amd/graphed.py ``prefetch``)."""

    def finish(self):
        if self.done is not None:
            return self.done
        self.event.synchronize()                             # the one host wait of the batch
        vals = self.meta_host.tolist()
        _pinned_meta.append(self.meta_host)
        self.done = _finish_graph(self, vals)
        return self.done


def build_graph(pos, batch, cutoff, max_num_neighbors=32, loop=False, triplets=True):
    """radius graph (+ CSR, + triplet lists) for a batch of molecules.

    pos f32 [N,3] (cuda), batch i64 [N] sorted.  Raises like the reference's dependency would on
    malformed input (RuntimeError)."""
    return start_graph(pos, batch, cutoff, max_num_neighbors, loop, triplets).finish()


def start_graph(pos, batch, cutoff, max_num_neighbors=32, loop=False, triplets=True):
    """stage 1 of ``build_graph`` without the host wait -> PendingGraph."""
    if not pos.is_cuda:
        raise _hip.Dig3dError('dig_amd runs on the GPU only (pos is not a cuda tensor); there is no CPU fallback')
    if pos.dtype != torch.float32 or pos.dim() != 2 or pos.size(1) != 3:
        raise RuntimeError(f'pos must be float32 [N,3], got {pos.dtype} {tuple(pos.shape)}')
    if batch is None:
        batch = torch.zeros(pos.size(0), dtype=torch.int64, device=pos.device)
    if batch.dtype != torch.int64 or batch.numel() != pos.size(0):
        raise RuntimeError('batch must be int64 [N]')
    posd = pos.detach().contiguous()
    batch = batch.contiguous()
    dev = pos.device
    N = pos.size(0)
    W = max_num_neighbors + (0 if loop else 1)
    slots = max(N * W, 1)
    i32 = dict(dtype=torch.int32, device=dev)
    g = MolGraph()
    g.N, g.batch = N, batch
    g_ptr = torch.empty(N + 2, **i32)
    nbr = torch.empty(slots, **i32)
    deg = torch.empty(max(N, 1), **i32)
    rowptr = torch.empty(N + 1, **i32)
    src = torch.empty(slots, **i32)
    dst = torch.empty(slots, **i32)
    cnt = torch.empty(slots, **i32)
    tptr = torch.empty(slots + 1, **i32)
    meta = torch.empty(8, dtype=torch.int64, device=dev)
    ws = torch.empty(slots // 4096 + 2, **i32)
    b32 = torch.empty(max(N, 1), **i32)[:N]          # the batch vector as int32, written by the pointer kernel
    st = _stream()
    pend = PendingGraph()
    pend.done = None
    if N == 0:                              # empty batch: nothing to launch
        g.B = g.E = g.T = 0
        g.ptr = torch.zeros(1, **i32)
        g.rowptr = torch.zeros(1, **i32)
        g.src = g.dst = g.col = g.kj = g.ji = torch.zeros(0, **i32)
        g.val, g.deg, g.batch32 = None, deg[:0], batch.to(torch.int32)
        g.tptr = torch.zeros(1, **i32)
        pend.done = g
        return pend
    # (B, E, T) reach the host through pinned memory the LAST kernel of the build writes itself (no copy command: the
    # framework's device->host copy was one more blit kernel in every step)
    pend.meta_host = _pinned_meta.pop() if _pinned_meta else torch.empty(8, dtype=torch.int64).pin_memory()
    call('dig3d_graph_build', ptr(posd), ptr(batch), N, float(cutoff), int(max_num_neighbors), int(bool(loop)),
         ptr(g_ptr), ptr(nbr), ptr(deg), ptr(rowptr), ptr(src), ptr(dst), ptr(cnt), ptr(tptr), ptr(meta), ptr(ws),
         int(bool(triplets)), ptr(b32), pend.meta_host.data_ptr(), st)
    g.batch32 = b32
    pend.event = torch.cuda.Event()
    pend.event.record()
    pend.g, pend.triplets, pend.i32 = g, triplets, i32
    pend.bufs = (g_ptr, rowptr, src, dst, deg, tptr, meta, posd, nbr, cnt, ws)     # keep stage-1 storage alive
    return pend


def _finish_graph(pend, vals):
    g, triplets, i32 = pend.g, pend.triplets, pend.i32
    g_ptr, rowptr, src, dst, deg, tptr = pend.bufs[:6]
    N = g.N
    st = _stream()
    B, E, T, _, _, _, _, err = vals
    if err & 1:
        raise RuntimeError('batch vector must be sorted ascending (torch_cluster.radius_graph requirement)')
    if err & 2:
        raise RuntimeError('batch ids must lie in [0, num_nodes)')
    g.B, g.E, g.T = int(B), int(E), int(T) if triplets else 0
    g.ptr = g_ptr[:g.B + 1]
    g.rowptr = rowptr
    g.src, g.dst = src[:g.E], dst[:g.E]
    g.col, g.val = g.src, None
    g.deg = deg[:N]
    if triplets:
        g.tptr = tptr[:g.E + 1]
        g.kj = torch.empty(max(g.T, 1), **i32)[:g.T]
        g.ji = torch.empty(max(g.T, 1), **i32)[:g.T]
        call('dig3d_graph_triplets_fill', ptr(rowptr), ptr(g.src), None, ptr(g.src), ptr(g.dst), ptr(g.tptr),
             g.E, ptr(g.kj), ptr(g.ji), st)
    pend.bufs = None
    return g


def graph_from_edge_index(edge_index, num_nodes, triplets=True):
    """MolGraph for a caller-supplied int64 edge_index (public ``xyz_to_dat`` path).  The CSR by
    (target, source) is derived with a stable sort exactly as torch_sparse.SparseTensor does
    (geometric_computing.py:27-28); triplets are ordered by the ORIGINAL edge id."""
    dev = edge_index.device
    j, i = edge_index[0].contiguous(), edge_index[1].contiguous()
    E = j.numel()
    g = MolGraph()
    g.N, g.E = int(num_nodes), E
    i32 = dict(dtype=torch.int32, device=dev)
    g.src = torch.empty(max(E, 1), **i32)[:E]
    g.dst = torch.empty(max(E, 1), **i32)[:E]
    st = _stream()
    if E:
        call('dig3d_cast_i64_i32', ptr(j), ptr(g.src), E, st)
        call('dig3d_cast_i64_i32', ptr(i), ptr(g.dst), E, st)
    key = i * num_nodes + j
    is_sorted = bool((key[1:] >= key[:-1]).all()) if E > 1 else True
    if is_sorted:
        col, val = g.src, None
    else:
        order = torch.argsort(key, stable=True)
        col, val = g.src[order].contiguous(), order.to(torch.int32)
    deg = torch.bincount(i, minlength=num_nodes).to(torch.int32)
    rowptr = torch.empty(num_nodes + 1, **i32)
    ws = torch.empty(max(num_nodes, E) // 4096 + 2, **i32)
    call('dig3d_scan_i32', ptr(deg), ptr(rowptr), int(num_nodes), None, ptr(ws), st)
    g.rowptr, g.col, g.val = rowptr, col, val
    g._sorted_edges = is_sorted
    if triplets:
        cnt = torch.empty(max(E, 1), **i32)
        tptr = torch.empty(E + 1, **i32)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        call('dig3d_graph_triplets_count', ptr(rowptr), ptr(col), ptr(g.src), ptr(g.dst), E, ptr(cnt), ptr(tptr),
             ptr(total), ptr(ws), st)
        g.T = int(total.item())
        g.tptr = tptr
        g.kj = torch.empty(max(g.T, 1), **i32)[:g.T]
        g.ji = torch.empty(max(g.T, 1), **i32)[:g.T]
        call('dig3d_graph_triplets_fill', ptr(rowptr), ptr(col), ptr(val), ptr(g.src), ptr(g.dst), ptr(tptr), E,
             ptr(g.kj), ptr(g.ji), st)
    return g
