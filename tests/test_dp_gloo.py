"""CPU, world_size 2 over gloo: the data-parallel plumbing of dig_amd/dp.py.

The engine's kernels need a GPU, the DP layer does not: it is one flat gradient bucket + one all-reduce, model
agnostic.  Checked here: (1) shards are disjoint, equal-sized and cover the set; (2) after ``allreduce`` every
rank's bucket equals the single-process gradient of the L1-mean loss on the concatenated batch; (3) ``.grad``
tensors are views of the reduced flat buffer after ``allreduce`` (one pack kernel, no unpack); (4) the
validation sums reduce to the global value."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 16), nn.SiLU(), nn.Linear(16, 1))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 6, generator=g), torch.randn(8, 1, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dig_amd import dp
    r, w = dp.init_from_env('gloo')
    assert (r, w) == (rank, world)
    x, y = _data()
    idx = dp.shard_indices(len(x), r, w)
    model = _model()
    bucket = dp.GradBucket(model)
    bucket.zero()
    loss = (model(x[idx]) - y[idx]).abs().mean()
    loss.backward()
    bucket.allreduce()
    flat1 = bucket.flat.clone()
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
    for p in model.parameters():                       # (3) .grad are views of the reduced buffer
        assert lo <= p.grad.data_ptr() < hi
    # second step: zero() drops the views, backward assigns fresh grads, allreduce packs again
    bucket.zero()
    assert all(p.grad is None for p in model.parameters())
    loss = (model(x[idx]) - y[idx]).abs().mean()
    loss.backward()
    bucket.allreduce()
    tot = dp.allreduce_scalar_sum(float(len(idx)), 'cpu')
    q.put((rank, idx, flat1, bucket.flat.clone(), tot))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gradient_bucket_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # (1) shards
    i0, i1 = res[0][1], res[1][1]
    assert len(i0) == len(i1) == 4 and sorted(i0 + i1) == list(range(8))
    # (2) reference: single process, whole batch
    x, y = _data()
    m = _model()
    (m(x) - y).abs().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    for _, _, flat1, flat2, tot in res:
        assert torch.allclose(flat1, ref, atol=1e-7), (flat1 - ref).abs().max()
        assert torch.allclose(flat2, ref, atol=1e-7)
        assert tot == 8.0
    assert torch.equal(res[0][2], res[1][2])           # bit-identical on both ranks


def test_single_process_is_a_noop():
    from dig_amd import dp
    assert dp.world_size() == 1 and dp.rank() == 0
    m = _model()
    b = dp.GradBucket(m)
    b.zero()
    x, y = _data()
    (m(x) - y).abs().mean().backward()
    before = [p.grad.clone() for p in m.parameters()]
    b.allreduce()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, m.parameters()))
    assert dp.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
