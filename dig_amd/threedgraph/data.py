"""Minimal molecule-batch plumbing (what run.py:53-55 gets from torch_geometric's DataLoader/Batch,
SURVEY.md A.7): concatenate per-molecule ``z, pos[, force, node_feature]`` along dim 0, stack ``y``, add the
sorted ``batch`` vector and ``ptr``.  Any object with attributes z/pos/y works as a sample (PyG ``Data``
included), so reference datasets plug in unchanged.
"""
from types import SimpleNamespace

import torch

_NODE_KEYS = ('z', 'pos', 'force', 'node_feature')


class MolBatch(SimpleNamespace):
    def to(self, device, non_blocking=False):
        out = MolBatch(**vars(self))
        for k, v in vars(self).items():
            if torch.is_tensor(v):
                setattr(out, k, v.to(device, non_blocking=non_blocking))
        return out

    def pin_memory(self):
        out = MolBatch(**vars(self))
        for k, v in vars(self).items():
            if torch.is_tensor(v):
                setattr(out, k, v.pin_memory())
        return out


def collate(samples):
    out = MolBatch()
    n = torch.tensor([int(s.z.size(0)) for s in samples], dtype=torch.int64)
    for k in _NODE_KEYS:
        vals = [getattr(s, k, None) for s in samples]
        if all(torch.is_tensor(v) for v in vals):
            setattr(out, k, torch.cat(vals, 0))
        elif k == 'node_feature':
            out.node_feature = None
    ys = [getattr(s, 'y', None) for s in samples]
    if all(v is not None for v in ys):
        out.y = torch.cat([torch.as_tensor(v).reshape(-1) for v in ys], 0)
    out.batch = torch.arange(len(samples), dtype=torch.int64).repeat_interleave(n)
    out.ptr = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
    out.ptr_list = out.ptr.tolist()          # host copy (dig_amd/graphed.py splits batches without a device read)
    out.num_graphs = len(samples)
    set_z_bounds(out)
    return out


def set_z_bounds(batch):
    """host-side (min, max) of the atomic numbers, read while ``z`` is still a host tensor: the engine's embedding kernel
    clamps its index instead of asserting like ``nn.Embedding`` (a device-side check would cost a synchronisation per
    forward), so the models validate these two Python ints instead (``check_z_bounds``)."""
    z = getattr(batch, 'z', None)
    if torch.is_tensor(z) and not z.is_cuda and z.numel():
        batch.z_bounds = (int(z.min()), int(z.max()))
    return batch


def check_z_bounds(batch, rows):
    """IndexError — what ``torch.nn.Embedding`` raises on the CPU (method/spherenet/spherenet.py:70: ``self.emb(z)``) — when a
    loader batch carries an atomic number outside the embedding table of ``rows`` rows.  Batches without ``z_bounds``
    (tensors handed over directly on the device) are not checked."""
    zb = getattr(batch, 'z_bounds', None)
    if zb is not None and (zb[0] < 0 or zb[1] >= rows):
        raise IndexError(f'index out of range in self: atomic numbers span [{zb[0]}, {zb[1]}], the embedding table has '
                         f'{rows} rows')


class _Positions(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class DataLoader(torch.utils.data.DataLoader):
    """torch_geometric.data.DataLoader(dataset, batch_size, shuffle) equivalent.  Datasets that store their
    molecules flat (dig_amd.threedgraph.dataset) are batched by ONE vectorised gather per batch
    (``dataset.collate_indices``) instead of per-sample Python objects."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop('collate_fn', None)
        self.molecules = dataset
        flat = hasattr(dataset, 'collate_indices')
        src = _Positions(len(dataset)) if flat else dataset
        fn = dataset.collate_indices if flat else collate
        if kwargs.get('batch_sampler') is not None:            # data-parallel plans (dig_amd/dp.py) bring their batches
            super().__init__(src, collate_fn=fn, **kwargs)
        else:
            super().__init__(src, batch_size, shuffle, collate_fn=fn, **kwargs)


# ---------------------------------------------------------------------------------------------------------------
# host -> device pipeline: pre-batched, pinned, double buffered
# ---------------------------------------------------------------------------------------------------------------
_DEVICE_KEYS = ('z', 'pos', 'batch', 'y', 'force', 'node_feature')


class DeviceLoader:
    """Wraps a (host) loader of molecule batches and yields them ON THE DEVICE, ``depth`` batches ahead of use:

      * a worker thread runs the collate (Python / index arithmetic) so it overlaps the GPU step;
      * every batch is packed into ONE pinned staging buffer (z, pos, batch, y[, force, node_feature] back to back,
        16-byte aligned) and crosses PCIe as ONE asynchronous copy on a dedicated copy stream;
      * ``depth`` staging/device slot pairs are recycled: the copy into a slot waits (stream-ordered, no host sync)
        for the compute stream to be done with the batch that lived there.

    The reference builds a PyG ``Batch`` per step on the host and calls ``.to(device)`` (run.py:53-55,123: pageable
    memory, several small synchronous copies) — at a 4 ms GPU step that is the critical path.  ``ptr_list`` (host copy
    of the graph pointer) travels with the batch: dig_amd/graphed.py needs sizes without a device read."""

    def __init__(self, loader, device, depth=4, lag=1):
        # lag: how many batches the consumer keeps alive BEHIND the one it was just handed (run.train holds the
        # current batch while it already asked for the next one: lag = 1); depth >= lag + 2 slots
        self.loader, self.device, self.lag = loader, torch.device(device), max(0, int(lag))
        self.depth = max(self.lag + 2, int(depth))
        self._slots = None

    def __len__(self):
        return len(self.loader)

    @property
    def batch_sampler(self):
        return getattr(self.loader, 'batch_sampler', None)

    @staticmethod
    def _layout(batch):
        """[(key, tensor, byte offset)], total bytes — every tensor's slice 16-byte aligned."""
        items, off = [], 0
        # the model inputs first (fixed order), then EVERY other tensor attribute of the batch (``ptr``, ProNet's
        # coords_* / side-chain embeddings, custom keys): what ``batch.to(device)`` of the reference moves (run.py:123)
        keys = [k for k in _DEVICE_KEYS if torch.is_tensor(getattr(batch, k, None))]
        keys += [k for k, v in vars(batch).items() if torch.is_tensor(v) and k not in _DEVICE_KEYS]
        for k in keys:
            t = getattr(batch, k).contiguous()
            items.append((k, t, off))
            off += (t.numel() * t.element_size() + 15) // 16 * 16
        return items, max(off, 16)

    def _slot(self, k, nbytes):
        if self._slots is None:
            self._slots = [None] * self.depth
        s = self._slots[k]
        if s is None or s['host'].numel() < nbytes:
            cap = max(nbytes * 5 // 4, 1 << 16)
            s = dict(host=torch.empty(cap, dtype=torch.uint8).pin_memory(),
                     dev=torch.empty(cap, dtype=torch.uint8, device=self.device), free=None, ready=None)
            # the device buffer comes from the caching allocator of the COMPUTE stream: the block may have belonged to a
            # tensor whose last kernels are still queued there (the host runs ahead of the GPU), and the allocator's
            # stream-order guarantee does not cover the copy stream that writes into it — so the first copy into a new
            # buffer waits for everything queued on the compute stream so far.  (Found in round 3: a graph array freed
            # after StaticGraph.load was recycled as a slot and overwritten under the pending load kernel.)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            s['free'] = ev
            self._slots[k] = s
        return s

    def _stage(self, batch, k, copy_stream):
        if getattr(batch, 'z_bounds', None) is None:
            set_z_bounds(batch)                       # (collate functions other than ours: still a host tensor here)
        items, nbytes = self._layout(batch)
        s = self._slot(k, nbytes)
        host = s['host']
        if s['ready'] is not None:
            # the previous async copy OUT of this pinned buffer must have finished before the host rewrites it (it is
            # ``depth`` batches old, so this wait is almost always free; without it only the device side was ordered)
            s['ready'].synchronize()
        for key, t, off in items:                                   # tiny memcpys into the pinned staging buffer
            n = t.numel() * t.element_size()
            host[off:off + n].copy_(t.reshape(-1).view(torch.uint8))
        with torch.cuda.stream(copy_stream):
            if s['free'] is not None:
                copy_stream.wait_event(s['free'])                   # the compute stream is done with this slot
            s['dev'][:nbytes].copy_(host[:nbytes], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(copy_stream)
            s['ready'] = ready
        out = MolBatch(**{k_: v for k_, v in vars(batch).items() if not torch.is_tensor(v)})
        for key, t, off in items:
            n = t.numel() * t.element_size()
            setattr(out, key, s['dev'][off:off + n].view(t.dtype).view(t.shape))
        if torch.is_tensor(getattr(batch, 'ptr', None)):
            out.ptr_list = getattr(batch, 'ptr_list', None) or batch.ptr.tolist()
        return out, ready, s

    def __iter__(self):
        import queue
        import threading
        if self.device.type != 'cuda':
            for b in self.loader:
                yield b.to(self.device)
            return
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def work():                       # collate only: no CUDA call ever leaves this thread (safe beside a capture)
            # the collate is a handful of tiny index operations: with torch's default intra-op pool (one OpenMP thread
            # per host core — 256 on the GPU box) every one of them pays a fork/join and the spinning pool starves the
            # thread that launches the GPU step: measured 15.1 ms per step through the loader against 2.6 ms on resident
            # batches (r03, profiles/r03a_bench_line_default.json); the setting is per calling thread
            torch.set_num_threads(1)
            try:
                for b in self.loader:
                    if stop.is_set():
                        return
                    q.put(b)
                q.put(None)
            except BaseException as e:    # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        copy_stream = torch.cuda.Stream(self.device)
        inflight = []                      # (batch on device, ready event, slot)
        handed = []                        # slots of the batches the consumer may still be using (newest last)
        k = 0
        done = False
        try:
            while True:
                cur = torch.cuda.current_stream(self.device)
                while len(handed) > self.lag:              # older than the consumer's look-behind: released once the
                    ev = torch.cuda.Event()                # work queued so far has run (stream-ordered, no host sync)
                    ev.record(cur)
                    handed.pop(0)['free'] = ev
                while not done and len(inflight) + len(handed) < self.depth - 1:
                    item = q.get()
                    if item is None:
                        done = True
                        break
                    if isinstance(item, BaseException):
                        raise item
                    inflight.append(self._stage(item, k % self.depth, copy_stream))
                    k += 1
                if not inflight:
                    break
                out, ready, slot = inflight.pop(0)
                cur.wait_event(ready)
                handed.append(slot)
                yield out
        finally:
            stop.set()
            while not q.empty():
                try:
                    q.get_nowait()
                except Exception:
                    break
