"""Data parallelism for dig.threedgraph: molecules shard by graph, ONE collective per step.

The reference is single-device (no DDP/NCCL anywhere).  Molecule batches never share
edges (radius_graph is restricted to a graph), so the only exchange is the gradient sum: one flat float32
bucket (0.35-15 MB for these models) all-reduced with RCCL over xGMI (backend "nccl" on ROCm), or gloo on CPU
in the tests.  At these sizes the ring is latency- not bandwidth-bound, so a single bucket and no overlap is
the right shape: one pack kernel, one collective, gradients become views of the reduced buffer.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend=None):
    """torchrun-style init (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  No-op for a single process."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    force = os.environ.get('DIG3D_FORCE_DIST') == '1'      # exercise the collective path with a 1-rank group
    if (ws <= 1 and not force) or is_dist():
        return rank(), world_size()
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend)
    return rank(), world_size()


class GradBucket:
    """One flat float32 buffer for the single gradient all-reduce of a step.

    ``zero()`` drops the ``.grad`` tensors, so autograd ASSIGNS fresh gradients during backward instead of
    accumulating into zero-filled storage (that costs one add kernel per parameter — 135 launches per SphereNet
    step).  ``allreduce()`` packs them with ONE concatenation kernel, all-reduces the flat buffer and re-points
    every ``p.grad`` at its slice of the reduced buffer (views: no unpack copy).  With a single process both
    calls launch nothing."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = None

    def zero(self):
        for p in self.params:
            p.grad = None

    def allreduce_flat(self, flat):
        """all-reduce a gradient buffer that is ALREADY flat and pre-scaled (dig_amd/graphed.py produces it inside
        the HIP graph): the whole data-parallel exchange of a step is this one collective."""
        if is_dist():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    def allreduce_flat_start(self, flat):
        """the same collective, asynchronous: enqueued behind the work already on the current stream (the replay) and
        running beside whatever the caller enqueues next (the next batch's graph build); ``allreduce_flat_finish`` makes
        the current stream wait for it before the optimizer reads the buffer."""
        self._work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True) if is_dist() else None

    def allreduce_flat_finish(self):
        work, self._work = getattr(self, '_work', None), None
        if work is not None:
            work.wait()

    def allreduce(self, scale=None):
        """sum over ranks of (scale x local gradient).  ``scale`` = B_local / B_global is this rank's share of the
        global batch: the sum is then the gradient of the mean-reduced loss over the concatenated
        batch also when the ranks hold different numbers of graphs; default 1/world (equal shards)."""
        if not is_dist():
            return
        grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        self.flat = torch.cat(grads)
        if scale is None:
            scale = 1.0 / world_size()
        if scale != 1.0:
            self.flat.mul_(scale)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()


def shard_indices(n, rank_, world, shuffle_seed=None, epoch=0, drop_tail=True):
    """Contiguous shard of range(n).  ``drop_tail=True``: every rank gets n // world items (training shards of equal
    size).  ``drop_tail=False``: ragged shards that COVER the set (the first n % world ranks get one more) — what
    validation / test must use so that the all-reduced MAE is the MAE of the whole set (run.py:171-180)."""
    idx = torch.arange(n)
    if shuffle_seed is not None:
        g = torch.Generator().manual_seed(shuffle_seed + epoch)
        idx = idx[torch.randperm(n, generator=g)]
    if world <= 1:
        return idx.tolist()
    if drop_tail:
        per = n // world
        return idx[rank_ * per:(rank_ + 1) * per].tolist()
    per, rem = divmod(n, world)
    lo = rank_ * per + min(rank_, rem)
    return idx[lo:lo + per + (1 if rank_ < rem else 0)].tolist()


def molecule_cost(num_atoms, max_num_neighbors=32):
    """Estimated work of one molecule: the triplet count n * deg^2 with deg = min(n - 1, cap) dominates every model of
    the path (SURVEY.md §8e "balance by ... estimated sum deg^2")."""
    n = torch.as_tensor(num_atoms, dtype=torch.float64)
    deg = torch.clamp(n - 1, min=1, max=max_num_neighbors)
    return n * deg * deg


class BalancedBatchSampler(torch.utils.data.Sampler):
    """Batch sampler of a data-parallel epoch.  Every rank runs the SAME deterministic plan (shared seed + epoch):
    the shuffled set is cut into global batches of ``batch_size * world`` graphs; inside a global batch the graphs are
    sorted by estimated cost and dealt to the ranks in snake order (0..w-1, w-1..0, ...), so every rank steps the same
    number of graphs with nearly the same sum of costs — the all-reduce waits for the slowest rank, and on 40-120-atom
    systems a plain contiguous split leaves up to 2x imbalance.  A trailing global batch with fewer graphs than ranks
    is dropped (a rank without graphs could not take part in the step); otherwise nothing is dropped: the last batch
    is ragged and ``weights[k]`` = B_local / B_global of step k is what the local gradient must be scaled by.

    ``costs`` None -> unit costs (plain round-robin deal)."""

    def __init__(self, n, batch_size, rank_, world, costs=None, shuffle=True, seed=0):
        self.n, self.bs, self.rank, self.world = int(n), int(batch_size), int(rank_), int(world)
        self.costs = None if costs is None else torch.as_tensor(costs, dtype=torch.float64)
        self.shuffle, self.seed, self.epoch = shuffle, int(seed), 0
        self.weights = []
        self._plan_cache = None

    def set_epoch(self, epoch):
        self.epoch = int(epoch)
        self._plan_cache = None

    def plan(self):
        """-> (batches of this rank, weights B_local / B_global), identical arithmetic on every rank."""
        if self._plan_cache is not None:
            return self._plan_cache
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        gb = self.bs * self.world
        batches, weights = [], []
        for a in range(0, self.n, gb):
            ids = order[a:a + gb]
            if ids.numel() < self.world:
                break
            if self.costs is not None:
                ids = ids[torch.argsort(self.costs[ids], descending=True, stable=True)]
            pos = torch.arange(ids.numel())
            lap, col = pos // self.world, pos % self.world
            owner = torch.where(lap % 2 == 0, col, self.world - 1 - col)
            mine = ids[owner == self.rank]
            batches.append(mine.tolist())
            weights.append(mine.numel() / ids.numel())
        self._plan_cache = (batches, weights)
        return self._plan_cache

    def __iter__(self):
        batches, self.weights = self.plan()
        self.epoch += 1                   # next epoch reshuffles (DistributedSampler.set_epoch semantics, automatic)
        self._plan_cache = None
        return iter(batches)

    def __len__(self):
        return len(self.plan()[0])


class ListBatchSampler(torch.utils.data.Sampler):
    """fixed index list cut into consecutive batches (validation / test shards)."""

    def __init__(self, indices, batch_size):
        self.indices, self.bs = list(indices), int(batch_size)

    def __iter__(self):
        return iter([self.indices[a:a + self.bs] for a in range(0, len(self.indices), self.bs)])

    def __len__(self):
        return -(-len(self.indices) // self.bs)


def broadcast_parameters(model, optimizer=None, src=0):
    """Make every rank start from rank ``src``'s weights (the reference is single-process; DP replicas built with
    different RNG state would otherwise train apart silently).  With FlatAdam the parameters already live in one flat
    buffer: ONE broadcast; otherwise one per tensor.  Buffers are broadcast as well."""
    if not is_dist():
        return
    done = set()
    if optimizer is not None:
        for group in optimizer.param_groups:
            fl = group.get('_flat')
            if fl is not None:
                dist.broadcast(fl['param'], src)
                done.update(id(p) for p in fl['ps'])
    for p in model.parameters():
        if id(p) not in done:
            dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b.data, src)


def allreduce_scalar_sum(x, device):
    if not is_dist():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
