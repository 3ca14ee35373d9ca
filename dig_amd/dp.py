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

    def allreduce(self, scale=None):
        """sum over ranks (x scale).  With equal shard sizes scale = 1/world reproduces the single-process
        L1 'mean' loss gradient on the concatenated batch."""
        if not is_dist():
            return
        grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        self.flat = torch.cat(grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if scale is None:
            scale = 1.0 / world_size()
        if scale != 1.0:
            self.flat.mul_(scale)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()


def shard_indices(n, rank_, world, shuffle_seed=None, epoch=0):
    """Contiguous-by-stride shard of range(n) (drop-tail so every rank gets the same count)."""
    idx = torch.arange(n)
    if shuffle_seed is not None:
        g = torch.Generator().manual_seed(shuffle_seed + epoch)
        idx = idx[torch.randperm(n, generator=g)]
    per = n // world
    return idx[rank_ * per:(rank_ + 1) * per].tolist() if world > 1 else idx.tolist()


def allreduce_scalar_sum(x, device):
    if not is_dist():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
