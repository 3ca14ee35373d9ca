"""HIP-graph execution of a training step over static-shape (bucketed) molecule batches.

At the reference's batch sizes (QM9, 32 molecules: N ~ 600 atoms, E ~ 10^4 edges, T ~ 10^5 triplets) a step is
several hundred short kernels; launched one by one from Python the step is bound by host launch latency and the
GPU idles between kernels.  Here forward + loss + backward of a model is captured ONCE per
shape bucket as a HIP graph and replayed with a single launch.

Static shapes without changing results:
  * every per-batch array (positions, edge list, CSRs, triplet lists, transposed CSRs) lives in a buffer of a bucket
    capacity (N_cap, E_cap, T_cap), the live counts sit in device memory (``cnt``);
  * segment kernels are CSR driven — the padded tail of every row-pointer array equals the total, so padded
    segments are empty and contribute exact zeros;
  * row gathers / geometry kernels take ``cnt`` and write exact zeros (or a harmless constant) for padded rows, so
    every gradient entering a padded row is exactly zero and weight-gradient reductions over the padded rows add
    0.0 — the graphed step returns bit-for-bit the gradients of the eager step on the unpadded batch
    (tests/test_gpu_models.py::test_graphed_step_equals_eager).
The radius graph itself (the only stage whose output SIZE is data dependent) runs eagerly before the replay: it
ends in the single device->host copy of (B, E, T) that also selects the bucket.
"""
import ctypes

import torch

from . import ops
from ._hip import call
from .graph import MolGraph, Seg, start_graph
from .optim import flat_layout


_BUCKET_BITS = 4        # capacity grid: 2^4 steps per octave


def bucket_cap(n, floor=64, bits=_BUCKET_BITS):
    """smallest multiple of 2^(k-bits) that is >= n, 2^k <= n < 2^(k+1): 2^bits steps per octave (16: <= 6.25 % padding),
    and a multiple of 64 rows (the dense kernels' row tile) from 1024 up."""
    n = max(int(n), floor)
    k = n.bit_length() - 1
    step = max(1 << max(k - bits, 0), 1)
    return -(-n // step) * step


def _words(t):
    return t.numel() * t.element_size() // 4


class StaticGraph(MolGraph):
    """A MolGraph whose arrays have bucket capacities; ``load`` refills them from an exact-size graph."""

    def __init__(self, n_cap, e_cap, t_cap, num_graphs, device, triplets=True):
        super().__init__()
        i32 = dict(dtype=torch.int32, device=device)
        self.triplets = bool(triplets)
        if not self.triplets:
            t_cap = 0
        self.N, self.E, self.T, self.B = n_cap, e_cap, t_cap, num_graphs
        self.cnt = torch.zeros(4, **i32)
        self.cnt_N, self.cnt_E, self.cnt_T = self.cnt[0:1], self.cnt[1:2], self.cnt[2:3]
        self.ptr = torch.zeros(num_graphs + 1, **i32)
        self.batch32 = torch.zeros(n_cap, **i32)
        self.rowptr = torch.zeros(n_cap + 1, **i32)
        self.src = torch.zeros(e_cap, **i32)
        self.dst = torch.zeros(e_cap, **i32)
        self.col, self.val = self.src, None
        self.tptr = torch.zeros(e_cap + 1, **i32)
        self.kj = torch.zeros(t_cap, **i32)
        self.ji = torch.zeros(t_cap, **i32)
        self._by_src = Seg(self.src, torch.zeros(n_cap + 1, **i32), torch.zeros(e_cap, **i32), n_cap, self.cnt_E)
        self._by_kj = Seg(self.kj, torch.zeros(e_cap + 1, **i32), torch.zeros(t_cap, **i32), e_cap, self.cnt_T)
        self.is_static_graph = True       # model.forward(sg) takes the prebuilt-graph route
        self.force = None                 # [n_cap, 3] target forces (energy_and_force batches)
        self.pos_leaf = None
        self.ones = None                  # grad_outputs of the force gradient ([B, out_channels] of ones), made once
        self.pos = torch.zeros(n_cap, 3, dtype=torch.float32, device=device)
        self.z = torch.zeros(n_cap, dtype=torch.int64, device=device)
        self.y = torch.zeros(num_graphs, dtype=torch.float32, device=device)
        self.node_feature = None          # [n_cap, D] extra per-node features (SphereNet use_extra_node_feature)

    def fits(self, g):
        return g.N <= self.N and g.E <= self.E and g.T <= self.T and g.B == self.B

    def load(self, g, z, pos, y, force=None, node_feature=None):
        """copy an exact-size graph (and the batch tensors) into the static buffers and pad the tails (row
        pointers with their totals, index arrays with 0) — ONE launch (csrc/graph.hip:k_pack_static)."""
        N, E, T = g.N, g.E, (g.T if self.triplets else 0)
        g.build_transposed(self.triplets)
        s = g.seg_src
        pos = pos.detach().contiguous()
        items = ((self.ptr, g.ptr, N), (self.batch32, g.batch32, 0), (self.rowptr, g.rowptr, E), (self.src, g.src, 0),
                 (self.dst, g.dst, 0), (self._by_src.kptr, s.kptr, E), (self._by_src.perm, s.perm, 0),
                 (self.pos, pos, 0), (self.z, z.contiguous(), 0), (self.y, y.contiguous(), 0))
        if self.triplets:
            k = g.seg_kj
            items = items + ((self.tptr, g.tptr, T), (self.kj, g.kj, 0), (self.ji, g.ji, 0),
                             (self._by_kj.kptr, k.kptr, T), (self._by_kj.perm, k.perm, 0))
        if force is not None:
            if self.force is None:
                self.force = torch.zeros_like(self.pos)
            items = items + ((self.force, force.contiguous(), 0),)
        if node_feature is not None:
            if self.node_feature is None:
                self.node_feature = torch.zeros(self.N, node_feature.size(1), dtype=torch.float32, device=self.pos.device)
            items = items + ((self.node_feature, node_feature.contiguous(), 0),)
        n = len(items)
        PP, IA, UA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_uint32 * n
        keep = [it[1] for it in items]                      # sources stay referenced until the launch is enqueued
        call('dig3d_pack_static',
             ctypes.cast(PP(*[it[1].data_ptr() if it[1].numel() else None for it in items]), ctypes.c_void_p),
             ctypes.cast(PP(*[it[0].data_ptr() for it in items]), ctypes.c_void_p),
             ctypes.cast(IA(*[_words(it[1]) for it in items]), ctypes.c_void_p),
             ctypes.cast(IA(*[_words(it[0]) for it in items]), ctypes.c_void_p),
             ctypes.cast(UA(*[it[2] for it in items]), ctypes.c_void_p), n,
             ctypes.cast((ctypes.c_int * 4)(N, E, T, 0), ctypes.c_void_p), self.cnt.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        del keep


class _Entry:
    __slots__ = ('sg', 'graph', 'loss', 'out', 'grads', 'flat', 'nbytes')


def l1_energy_loss(out, y):
    """run.py:127 with torch.nn.L1Loss (mean): one kernel forward, one backward (ops.l1_mean)."""
    return ops.l1_mean(out, y.unsqueeze(1))


class GraphedStep:
    """``loss = stepper(batch)`` == ``loss = loss_fn(model(batch), batch.y); loss.backward()`` with ``p.grad`` set,
    executed as one HIP-graph replay per step (plus the eager radius-graph prologue).  Models: SphereNet /
    DimeNetPP, energy only or energy_and_force (the double backward is captured whole; SphereNet's data-dependent
    torsion arg-min CSR is rebuilt by device-side kernels inside the graph)."""

    def __init__(self, model, loss_fn=l1_energy_loss, max_entries=32, grad_scale=1.0, force_loss=None, p=100.0):
        self.forces = bool(getattr(model, 'energy_and_force', False))
        # torch.nn.L1Loss() for energies AND forces (every reference example; run.py:49,127-129): the whole loss is one kernel
        self.l1_both = (loss_fn is l1_energy_loss or getattr(loss_fn, 'is_l1_mean', False)) and (
            force_loss is None or (isinstance(force_loss, torch.nn.L1Loss) and force_loss.reduction == 'mean'))
        # run.py:126-131: loss = loss_func(E) + p * loss_func(F); any mean-reduced elementwise loss works (the padded
        # rows carry zero force and zero target, the mean is rescaled to the live atom count)
        self.force_loss = force_loss or (lambda f, t: (f - t).abs().mean())
        self.p = float(p)
        self.model, self.loss_fn = model, loss_fn
        self.triplets = bool(getattr(model, 'needs_triplets', True))      # SchNet: edges only
        self.named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in self.named]
        self.entries = {}
        self.max_entries = max_entries
        self.captures = 0
        # data parallelism: gradients are produced pre-scaled (1/world) into ONE flat buffer inside the graph, so a
        # step is replay -> all_reduce(stepper.flat) -> optimizer.step() with no per-parameter host work
        self.grad_scale = float(grad_scale)
        # per-step scale (data parallelism with ragged last batches: B_local / B_global, dig_amd/dp.py) lives in a
        # device scalar read by the captured graph, so changing it needs no re-capture
        self.scale_t = None
        self._scale_val = None
        self.extra = bool(getattr(model, 'use_extra_node_feature', False))
        self._seed, self._seed_val = None, None
        self.flat = None
        self._bound = None
        self._pending = None
        emb = next((m for m in model.modules() if isinstance(m, torch.nn.Embedding)), None)
        self._emb_rows = emb.num_embeddings if emb is not None else None
        self.disabled = False
        # strict: a failed capture raises instead of degrading to kernel-by-kernel launches (benchmarks: a number
        # must never be reported under a replay label it did not earn)
        self.strict = False
        self.min_caps = (0, 0, 0)          # lower bounds for the bucket capacities (tests; coarse bucketing)

    def _fields(self, batch):
        """(z, pos, batch vector, y, force or None, node_feature or None) of a loader batch."""
        if self._emb_rows is not None:
            from .threedgraph.data import check_z_bounds
            check_z_bounds(batch, self._emb_rows)      # (the replayed forward never sees the loader batch itself)
        frc = getattr(batch, 'force', None) if self.forces else None
        nf = getattr(batch, 'node_feature', None) if self.extra else None
        return batch.z, batch.pos, batch.batch, batch.y, frc, nf

    def _backward_seed(self, device):
        """the gradient scale (1 / world, or B_local / B_global of a ragged data-parallel step) enters as the SEED of the
        backward pass — a device scalar — instead of as extra multiply nodes on the loss (and their backward kernels)"""
        if self.scale_t is not None:
            return self.scale_t.view(())
        # ONE seed tensor for the life of the stepper: captured graphs hold its ADDRESS, so a changed grad_scale is
        # written in place (a fresh tensor would hand the old block back to the allocator under every earlier capture)
        if self._seed is None:
            self._seed = torch.empty((), dtype=torch.float32, device=device)
            self._seed_val = None
        if self._seed_val != self.grad_scale:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('grad_scale changed inside a capture: set it before the step')
            self._seed.fill_(self.grad_scale)
            self._seed_val = self.grad_scale
        return self._seed

    def _run(self, sg):
        # The captured forward runs on fresh leaf ALIASES of the parameters (same storage, new autograd identity).
        # A parameter's AccumulateGrad node carries the stream of the forward that created it and stays alive while
        # any older autograd graph of that parameter is referenced (e.g. the loss of a previous eager step); the
        # autograd engine would switch to that stream inside the capture and the capture never re-joins
        # (hipStreamEndCapture then faults).  Aliases get their accumulators inside the capture.
        aliases = {n: p.detach().requires_grad_() for n, p in self.named}
        if self.forces:
            sg.pos_leaf = sg.pos.detach().requires_grad_()
        out = torch.func.functional_call(self.model, aliases, (sg,))
        seed = self._backward_seed(sg.pos.device)
        if self.forces and self.l1_both and out.is_cuda and out.dtype == torch.float32:
            # run.py:126-131 — force = -d sum(out) / d pos with create_graph, loss = L1(E) + p L1(F) — the loss as ONE launch
            # that also writes both gradients (the seed is known); the ones of grad_outputs live with the static batch
            if sg.ones is None or sg.ones.shape != out.shape:
                sg.ones = torch.ones_like(out)
            from . import diffops
            with diffops.force_gradient_scope():
                gpos = torch.autograd.grad(out, sg.pos_leaf, sg.ones, create_graph=True, retain_graph=True)[0]
            with ops.known_loss_seed(seed):
                loss = ops.ef_l1_loss(out, sg.y.unsqueeze(1), gpos, sg.force, sg.cnt_N, self.p)
            sg.pos_leaf = None
        else:
            with ops.known_loss_seed(seed):        # (the L1 loss writes its gradient in its forward launch)
                loss = self.loss_fn(out, sg.y)
            if self.forces:
                from . import diffops
                with diffops.force_gradient_scope():
                    force = -torch.autograd.grad(out, sg.pos_leaf, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
                loss = loss + self.p * self.force_loss(force, sg.force) * (float(sg.N) / sg.cnt_N.to(torch.float32)).squeeze()
                sg.pos_leaf = None
        # all weight-gradient partials of the step reduced by ONE launch (+ one accumulating launch for the weights
        # that enter the force path's graph twice: forward node and double-backward node)
        with ops.deferred_reductions() as red:
            grads = torch.autograd.grad(loss, list(aliases.values()), grad_outputs=seed, allow_unused=True)
        red.flush()
        # one flat, contiguous gradient buffer (a single pack kernel inside the graph)
        # (layout of dig_amd.optim.flat_layout — every parameter's slice 16-byte aligned — so FlatAdam consumes the
        # buffer in place)
        offs, total = flat_layout(self.params)
        flat = torch.empty(total, dtype=torch.float32, device=self.params[0].device)
        srcs = [None if gr is None else (gr if gr.is_contiguous() else gr.contiguous()).float() for gr in grads]
        n = len(srcs)
        IA = ctypes.c_int * n
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        call('dig3d_pack_flat', n, cast((ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in srcs])),
             cast(IA(*[p.numel() for p in self.params])), cast(IA(*[(p.numel() + 3) // 4 * 4 for p in self.params])),
             cast(IA(*offs)), flat.data_ptr(), torch.cuda.current_stream().cuda_stream)
        # p.grad are views of the flat buffer, laid out like their parameters
        views = [flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, offs)]
        return out, loss, flat, views

    def _capture(self, cap, g, fields):
        z, pos, _, y, frc, nf = fields
        mem0 = (torch.cuda.memory_allocated(pos.device), torch.cuda.memory_reserved(pos.device))
        sg = StaticGraph(cap[0], cap[1], cap[2], g.B, pos.device, triplets=self.triplets)
        sg.load(g, z, pos, y, frc, nf)
        # warm-up on a side stream (lazy allocations, library workspaces), gradients discarded
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._run(sg)
        torch.cuda.current_stream().wait_stream(s)
        e = _Entry()
        e.sg = sg
        e.graph = torch.cuda.CUDAGraph()
        # with a process group alive, RCCL's watchdog thread polls its work events (hipEventQuery) every 100 ms: under the
        # default 'global' capture mode such a call from ANOTHER thread invalidates the capture (a rare abort of the
        # single-rank RCCL test); 'thread_local' restricts the check to this thread
        import torch.distributed as dist
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'
        with torch.cuda.graph(e.graph, capture_error_mode=mode):
            e.out, e.loss, e.flat, e.grads = self._run(sg)
        self.captures += 1
        # what this size class keeps alive (static buffers + the graph's private pool): the eviction budget of __call__
        e.nbytes = max(torch.cuda.memory_allocated(pos.device) - mem0[0], torch.cuda.memory_reserved(pos.device) - mem0[1], 0)
        return e

    # ---- the NEXT batch's graph build, queued around the replay of the current step --------------------------------------
    # The build is ~25 launches of a few microseconds each (radius search, scans, triplet lists, two transposed CSRs).
    # Stage 1 (everything up to the (B, E, T) read-back) is enqueued BEFORE the replay is launched, so the sizes reach the
    # host while the GPU is inside the replay; stage 2 (the size-dependent rest) is enqueued right behind the replay.  The
    # next ``__call__`` then finds a finished graph and launches the refill + replay at once: the GPU never waits for the
    # host between steps (r03 same-box A/B: 2.011 -> 1.982 ms config 2, 1.068 -> 0.992 ms config 1).  Running the build on
    # a SIDE stream beside the replay was measured too and is slower than queueing it (2.011 vs 1.982 ms): the small
    # kernels take CUs from the replay's latency-bound kernels and the cross-stream events cost more than they hide.
    def _prefetch_stage1(self, batch):
        f = self._fields(batch)
        self._pending = [batch, f, start_graph(f[1], f[2], self.model.cutoff, triplets=self.triplets), None]

    def _prefetch_stage2(self):
        batch, f, pend, _ = self._pending
        g = pend.finish()                              # host waits for (B, E, T) only; the replay keeps the GPU busy
        g.build_transposed(self.triplets)              # transposed CSRs (StaticGraph.load reads them), one set of launches
        self._pending[3] = g

    def prefetch(self, batch):
        """build the graph of the NEXT batch now (both stages) — for callers that do not go through
        ``__call__(batch, prefetch=...)``."""
        self._prefetch_stage1(batch)
        self._prefetch_stage2()

    def class_key(self, g):
        """size class of a built graph: (batch size, edge bucket, triplet bucket, node bucket) — one HIP graph each"""
        return (g.B, bucket_cap(max(g.E, self.min_caps[1]), 1024), bucket_cap(max(g.T, self.min_caps[2]), 4096, bits=3),
                bucket_cap(max(g.N, self.min_caps[0]), bits=3))

    # ---- captures taken BEFORE the first step ----------------------------------------------------------------------------
    # A capture costs ~1 s.  Alone that is a one-off; data parallel it is a stall of EVERY rank (the other ranks wait at
    # the all-reduce), and each rank meets its size classes at different steps: 8 ranks x ~20 classes of a QM9-like epoch
    # would serialise into minutes of first-epoch stalls.  The sampler's plan is deterministic, so the trainer walks the
    # first epoch's batches once (graph builds only), the ranks exchange their class keys and every rank captures the
    # UNION up front, all at the same time (dig_amd/threedgraph/method/run.py).
    def scan_classes(self, batches):
        """-> {key: [count, a batch of that class]} for an iterable of loader batches (radius-graph builds only)"""
        seen = {}
        for b in batches:
            f = self._fields(b)
            g = start_graph(f[1], f[2], self.model.cutoff, triplets=self.triplets).finish()
            k = self.class_key(g)
            if k in seen:
                seen[k][0] += 1
            else:
                # a private copy: loader batches are views of recycled device slots (threedgraph/data.py DeviceLoader)
                import copy
                kept = copy.copy(b)
                for name, v in list(vars(b).items()):
                    if torch.is_tensor(v):
                        setattr(kept, name, v.clone())
                seen[k] = [1, kept]
        return seen

    def precapture(self, seen, keys=None):
        """capture the size classes ``keys`` (default: those of ``seen``), most frequent first, up to the entry bound;
        a class this rank has no batch of is captured on the largest local batch that fits its capacities (the static
        buffers are padded anyway).  -> number of captures made."""
        if self.disabled:
            return 0
        counts = {k: v[0] for k, v in seen.items()}
        if keys is None:
            keys = counts
        order = sorted(keys, key=lambda k: (-keys[k] if isinstance(keys, dict) else 0, k))
        made = 0
        dev = self.params[0].device
        # the same bounds __call__ keeps: max_entries and a quarter of the device memory (a 77k-edge OC20-like class holds
        # ~10 GB).  The most frequent classes are taken until the budget is reached; the rest is captured (or covered by a
        # larger graph) when its first batch arrives.
        budget = torch.cuda.get_device_properties(dev).total_memory // 4
        for key in order:
            if key in self.entries or len(self.entries) + 1 > self.max_entries:
                continue
            if sum(v.nbytes for v in self.entries.values()) > budget:
                break
            fit = [(k, v[1]) for k, v in seen.items() if k[0] == key[0] and k[1] <= key[1] and k[2] <= key[2] and k[3] <= key[3]]
            if not fit:
                continue
            _, batch = max(fit, key=lambda kv: kv[0][1:])
            f = self._fields(batch)
            g = start_graph(f[1], f[2], self.model.cutoff, triplets=self.triplets).finish()
            try:
                self.entries[key] = self._capture((key[3], key[1], key[2]), g, f)
            except RuntimeError as ex:              # out of memory, or another thread touched the device mid-capture
                if self.strict:
                    raise
                import warnings
                warnings.warn(f'HIP-graph pre-capture of size class {key} failed ({str(ex).splitlines()[0]}); '
                              f'{made} classes captured, the rest is captured on demand')
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                break
            made += 1
        return made

    def _eager(self, batch):
        """kernel-by-kernel step with the same contract (loss, p.grad views of self.flat) — used if a capture fails."""
        out = self.model(batch)
        loss = self.loss_fn(out, batch.y)
        if self.forces:
            from . import diffops
            with diffops.force_gradient_scope():
                force = -torch.autograd.grad(out, batch.pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
            loss = loss + self.p * self.force_loss(force, batch.force)
        obj = loss if self.grad_scale == 1.0 else loss * self.grad_scale
        if self.scale_t is not None:
            obj = obj * self.scale_t.squeeze()
        grads = torch.autograd.grad(obj, self.params, allow_unused=True)
        offs, total = flat_layout(self.params)
        flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        for p, gr, off in zip(self.params, grads, offs):
            if gr is not None:
                flat[off:off + p.numel()].copy_(gr.reshape(-1))
            p.grad = flat[off:off + p.numel()].view_as(p)
        self.flat, self._bound = flat, None
        return loss.detach()

    def set_scale(self, value):
        """gradient scale of the NEXT steps, replacing ``grad_scale`` (a device scalar the captured graphs read)."""
        value = float(value)
        if self.scale_t is None:
            if self.entries:
                raise RuntimeError('set_scale must be called before the first capture')
            self.scale_t = torch.ones(1, dtype=torch.float32, device=self.params[0].device)
            self.grad_scale = 1.0
        if value != self._scale_val:
            self.scale_t.fill_(value)
            self._scale_val = value

    def __call__(self, batch, prefetch=None, after_replay=None):
        """``after_replay(flat)``: called between the replay and the rest of the next batch's graph build — data parallelism
        starts its (asynchronous) gradient all-reduce there, so the collective runs beside the ~70 us of build kernels
        instead of behind them."""
        if self.disabled:
            loss = self._eager(batch)
            if after_replay is not None:
                after_replay(self.flat)
            return loss
        pend, self._pending = self._pending, None
        if pend is not None and pend[0] is batch:
            if pend[3] is None:                    # stage 2 not run yet (prefetch issued outside __call__)
                self._pending = pend
                self._prefetch_stage2()
                pend, self._pending = self._pending, None
            fields, g = pend[1], pend[3]
        else:
            fields = self._fields(batch)           # eager: sizes are data dependent (one host wait)
            g = start_graph(fields[1], fields[2], self.model.cutoff, triplets=self.triplets).finish()
        # One graph per SIZE CLASS (batch size, edge bucket, triplet bucket): a batch replays the tightest graph that holds
        # it.  (One graph per batch size whose capacities only grew converged to the LARGEST batch of the data set: the
        # B = 32 QM9-like batches have 6.4k - 9.2k edges and 82k - 127k triplets, so the average step ran on ~10 % padding
        # rows in every dense and triplet kernel.)  Edges: 16 steps per octave; triplets and nodes: 8 (they follow the edges, a
        # finer grid only multiplies the classes).
        # Same-box A/B against the single growing graph: config 2 1.887 -> 1.750 ms, config 4 6.257 -> 6.079 ms, config 3
        # (fixed-size molecules) equal; through the host loader 0.936 -> 0.989 of the resident rate.
        # The number of graphs is bounded (max_entries, and a quarter of the device memory: a 77k-edge OC20-like class holds
        # ~10 GB) without cycling captures WITHIN a batch size: once the bound is reached a batch of a new class replays the
        # tightest existing graph that holds it, and if none does, the LARGEST graph of its batch size is replaced by one
        # that covers both (that envelope only grows, like the single graph of before) — a capture costs ~1 s, a data set
        # with more classes than fit must not re-capture them in turn.  (Across batch sizes the bound can still evict: when
        # no entry of the batch's size exists the oldest entry of another size makes room — a data set's one ragged last
        # batch costs one capture per epoch at the bound, two alternating ragged sizes would re-capture each other.)
        key = self.class_key(g)
        e = self.entries.pop(key, None)
        cap = None
        if e is not None and e.sg.fits(g):
            self.entries[key] = e                  # (dict order = recency, kept for inspection)
        else:
            cap = (key[3], key[1], key[2])
            e = None
            budget = torch.cuda.get_device_properties(fields[1].device).total_memory // 4
            if len(self.entries) + 1 > self.max_entries or sum(v.nbytes for v in self.entries.values()) > budget:
                same = [(k, v) for k, v in self.entries.items() if k[0] == g.B]
                fit = [(k, v) for k, v in same if v.sg.fits(g)]
                if fit:                            # the tightest graph that holds the batch: no capture
                    key, e = min(fit, key=lambda kv: (kv[1].sg.E, kv[1].sg.T))
                    cap = None
                else:                              # grow the envelope: the largest graph of this batch size makes room
                    victim = max(same, key=lambda kv: (kv[1].sg.E, kv[1].sg.T)) if same else next(iter(self.entries.items()))
                    if victim[0][0] == g.B:
                        old = victim[1].sg
                        cap = (max(cap[0], old.N), max(cap[1], old.E), max(cap[2], old.T))
                    self.entries.pop(victim[0])
                    del victim, same, fit
        if cap is None:
            z, pos, _, y, frc, nf = fields
            e.sg.load(g, z, pos, y, frc, nf)
        else:
            try:
                e = self._capture(cap, g, fields)
            except RuntimeError as ex:              # e.g. another thread touched the device during the capture
                if self.strict:
                    raise
                import traceback
                import warnings
                where = ''.join(traceback.format_tb(ex.__traceback__)[-4:])
                warnings.warn(f'HIP-graph capture failed ({str(ex).splitlines()[0]}); continuing with kernel-by-kernel '
                              f'launches.  Raised at:\n{where}')
                self.disabled = True
                torch.cuda.synchronize()
                loss = self._eager(batch)
                if after_replay is not None:
                    after_replay(self.flat)
                return loss
            self.entries[key] = e
        if self._seed is not None and self._seed_val != self.grad_scale:      # grad_scale changed since the captures
            self._seed.fill_(self.grad_scale)
            self._seed_val = self.grad_scale
        if prefetch is not None:
            self._prefetch_stage1(prefetch)        # enqueued before the replay: the sizes reach the host during it
        e.graph.replay()
        if self._bound is not e or any(p.grad is not gr for p, gr in zip(self.params, e.grads)):
            for p, gr in zip(self.params, e.grads):
                p.grad = gr
            self._bound = e
        self.flat = e.flat
        self.last = e
        if after_replay is not None:
            after_replay(e.flat)
        if prefetch is not None:
            self._prefetch_stage2()
        return e.loss
