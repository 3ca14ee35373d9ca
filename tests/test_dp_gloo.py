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


def _worker(rank, world, port, q, ev):
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
    # the training loop's form for an already flat, pre-scaled buffer: asynchronous start, finish before the optimizer
    pre = bucket.flat.clone() / world
    want = pre.clone()
    dist.all_reduce(want)
    bucket.allreduce_flat_start(pre)
    bucket.allreduce_flat_finish()
    assert torch.equal(pre, want)
    bucket.allreduce_flat_finish()                     # idempotent when nothing is in flight
    q.put((rank, idx, flat1, bucket.flat.clone(), tot))
    ev.wait(60)                 # tensors in the queue are handles served by THIS process: stay until they are read
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gradient_bucket_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q, ev = ctx.Queue(), ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, ev)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    ev.set()
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


def _worker_ragged(rank, world, port, q, ev):
    """unequal local batches (5 + 3 graphs), replicas initialised under DIFFERENT seeds: after
    broadcast_parameters + allreduce(scale = B_local / B_global) every rank holds rank 0's weights and the
    gradient of the mean loss over all 8 graphs."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dig_amd import dp
    dp.init_from_env('gloo')
    x, y = _data()
    torch.manual_seed(100 + rank)                      # replicas built under different RNG state
    model = nn.Sequential(nn.Linear(6, 16), nn.SiLU(), nn.Linear(16, 1))
    before = torch.cat([p.detach().reshape(-1).clone() for p in model.parameters()])
    dp.broadcast_parameters(model)
    after = torch.cat([p.detach().reshape(-1).clone() for p in model.parameters()])
    idx = list(range(5)) if rank == 0 else list(range(5, 8))
    bucket = dp.GradBucket(model)
    bucket.zero()
    (model(x[idx]) - y[idx]).abs().mean().backward()
    bucket.allreduce(scale=len(idx) / 8.0)
    # ragged validation shards: exact global MAE from all-reduced sums and counts
    vidx = dp.shard_indices(7, rank, world, drop_tail=False)
    s = dp.allreduce_scalar_sum(float(sum(vidx)), 'cpu')
    n = dp.allreduce_scalar_sum(float(len(vidx)), 'cpu')
    q.put((rank, before, after, bucket.flat.clone(), vidx, s, n))
    ev.wait(60)                 # tensors in the queue are handles served by THIS process: stay until they are read
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_unequal_shards_and_parameter_broadcast():
    world = 2
    ctx = mp.get_context('spawn')
    q, ev = ctx.Queue(), ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q, ev)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    ev.set()
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, b0, a0, f0, v0, s0, n0), (_, b1, a1, f1, v1, s1, n1) = res
    assert not torch.equal(b0, b1)                      # the replicas really started apart
    assert torch.equal(a0, b0) and torch.equal(a1, b0)  # ... and now both hold rank 0's weights
    x, y = _data()
    torch.manual_seed(100)
    m = nn.Sequential(nn.Linear(6, 16), nn.SiLU(), nn.Linear(16, 1))
    (m(x) - y).abs().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(f0, ref, atol=1e-7) and torch.equal(f0, f1)
    assert sorted(v0 + v1) == list(range(7)) and len(v0) == 4 and len(v1) == 3      # nothing dropped
    assert (s0, n0) == (21.0, 7.0) == (s1, n1)


def test_balanced_batch_sampler_plan():
    """one deterministic plan on every rank: covers the set, equal graph counts per step, ragged last batch weighted
    B_local / B_global, per-step cost spread far below a contiguous split, reshuffled per epoch."""
    from dig_amd import dp
    g = torch.Generator().manual_seed(3)
    n_atoms = torch.randint(40, 121, (1003,), generator=g)
    costs = dp.molecule_cost(n_atoms)
    world, bs = 8, 32
    samplers = [dp.BalancedBatchSampler(1003, bs, r, world, costs, seed=5) for r in range(world)]
    plans = [s.plan() for s in samplers]
    steps = len(plans[0][0])
    assert all(len(p[0]) == steps for p in plans) and steps == 4              # 3 full global batches + ragged tail
    seen = []
    spread_bal, spread_naive = [], []
    for k in range(steps):
        sizes = [len(p[0][k]) for p in plans]
        assert abs(sum(p[1][k] for p in plans) - 1.0) < 1e-12                # weights of a step sum to one
        for p, sz in zip(plans, sizes):
            assert abs(p[1][k] - sz / sum(sizes)) < 1e-12
        if k < steps - 1:
            assert sizes == [bs] * world
            c = torch.tensor([costs[p[0][k]].sum().item() for p in plans])
            spread_bal.append((c.max() / c.mean()).item())
            allids = torch.tensor(sum((p[0][k] for p in plans), []))
            naive = costs[allids.sort().values].view(world, bs).sum(1)     # contiguous split of the same graphs
            spread_naive.append((naive.max() / naive.mean()).item())
        else:
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
        seen += sum((p[0][k] for p in plans), [])
    assert sorted(seen) == list(range(1003))                                 # nothing dropped, nothing repeated
    assert max(spread_bal) < 1.02 and max(spread_bal) < min(spread_naive)
    first = list(iter(samplers[0]))
    second = list(iter(samplers[0]))                                         # next epoch: another permutation
    assert first != second and first == plans[0][0]
    # fewer graphs than ranks in the tail: that tail is dropped (a rank without graphs cannot step)
    s = dp.BalancedBatchSampler(2 * 8 * 4 + 3, 4, 0, 8, None, shuffle=False)
    assert len(s) == 2
    assert dp.shard_indices(10, 1, 4, drop_tail=False) == [3, 4, 5] and dp.shard_indices(10, 3, 4, drop_tail=False) == [8, 9]
    assert dp.shard_indices(10, 3, 4) == [6, 7]


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


# ------------------------------------------------------------------------------------------------------------------
# run.run's data-parallel branch end to end (the per-step weights were read before the
# sampler had produced them).  The engine's models need a GPU; the TRAINER does not: a CPU stand-in with the models'
# interface (forward(batch_data) -> [B, 1] from z / pos / batch) goes through DataLoader -> BalancedBatchSampler.plan
# -> GradBucket.allreduce(scale = B_local / B_global) -> Adam -> ragged validation shards -> all-reduced MAE.
# ------------------------------------------------------------------------------------------------------------------
class _StandIn(nn.Module):
    """per-atom embedding + distance-to-centroid feature -> MLP -> sum over the atoms of a graph."""

    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(10, 8)
        self.mlp = nn.Sequential(nn.Linear(9, 16), nn.SiLU(), nn.Linear(16, 1))

    def forward(self, b):
        B = int(b.y.numel())
        n = torch.zeros(B).index_add_(0, b.batch, torch.ones(b.z.numel()))
        c = torch.zeros(B, 3).index_add_(0, b.batch, b.pos) / n[:, None]
        d = (b.pos - c[b.batch]).norm(dim=1, keepdim=True)
        h = self.mlp(torch.cat([self.emb(b.z), d], 1))
        return torch.zeros(B, 1).index_add_(0, b.batch, h)


def _mols(n, seed):
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        k = int(torch.randint(3, 9, (1,), generator=g))
        out.append(SimpleNamespace(z=torch.randint(1, 10, (k,), generator=g), pos=torch.randn(k, 3, generator=g),
                                   y=torch.randn(1, generator=g)))
    return out


N_TRAIN, BS = 23, 4          # global batches of 8: 8 + 8 + 7 graphs -> the last one is ragged (4 + 3)


def _worker_run(rank, world, port, q, ev):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dig_amd import dp
    from dig_amd.threedgraph.method.run import run
    from dig_amd.threedgraph.evaluation import ThreeDEvaluator
    dp.init_from_env('gloo')
    torch.manual_seed(100 + rank)                      # replicas start apart: run.run must broadcast rank 0's weights
    model = _StandIn()
    r = run()
    r.run(torch.device('cpu'), _mols(N_TRAIN, 1), _mols(7, 2), _mols(5, 3), model, nn.L1Loss(), ThreeDEvaluator(),
          epochs=2, batch_size=BS, vt_batch_size=2, lr=1e-2)
    # (plain lists: a tensor in an mp queue is a shared-memory handle that dies with this process)
    q.put((rank, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).tolist(), r.best_valid, r.best_test))
    ev.wait(60)                 # tensors in the queue are handles served by THIS process: stay until they are read
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_run_api_matches_single_process_on_global_batches():
    from dig_amd import dp
    from dig_amd.threedgraph.data import collate
    world = 2
    ctx = mp.get_context('spawn')
    q, ev = ctx.Queue(), ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_worker_run, args=(r, world, port, q, ev)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    ev.set()
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single process: the same plan's GLOBAL batches (union of both ranks' shares), plain Adam on the mean loss
    train, valid, test = _mols(N_TRAIN, 1), _mols(7, 2), _mols(5, 3)
    torch.manual_seed(100)
    model = _StandIn()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    n_at = torch.tensor([m.z.numel() for m in train])
    samplers = [dp.BalancedBatchSampler(N_TRAIN, BS, r, world, dp.molecule_cost(n_at), shuffle=True, seed=0)
                for r in range(world)]
    ragged = 0
    for epoch in range(2):
        plans = [s.plan() for s in samplers]
        for k in range(len(plans[0][0])):
            ids = plans[0][0][k] + plans[1][0][k]
            ragged += len(plans[0][0][k]) != len(plans[1][0][k])
            assert abs(plans[0][1][k] + plans[1][1][k] - 1.0) < 1e-12
            b = collate([train[i] for i in ids])
            opt.zero_grad()
            (model(b) - b.y.unsqueeze(1)).abs().mean().backward()
            opt.step()
        for s in samplers:
            list(iter(s))                                   # advances the epoch like the trainer's loader does

        def mae(data):
            with torch.no_grad():
                b = collate(data)
                return (model(b) - b.y.unsqueeze(1)).abs().mean().item()
        v, t = mae(valid), mae(test)
        if epoch == 0 or v < best_v:
            best_v, best_t = v, t
    assert ragged == 2                                      # the ragged last batch of both epochs was exercised
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    w0, w1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    assert torch.equal(w0, w1)                              # replicas stayed bit-identical
    assert (w0 - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (w0 - ref).abs().max()
    for _, _, bv, bt in res:
        assert abs(bv - best_v) <= 1e-5 * max(1.0, abs(best_v)) and abs(bt - best_t) <= 1e-5 * max(1.0, abs(best_t))


class _FakeStepper:
    """stands in for dig_amd.graphed.GraphedStep on the CPU: a batch's 'size class' is (graphs, atoms rounded up to 8)"""

    def __init__(self):
        self.captured = []

    def scan_classes(self, batches):
        seen = {}
        for b in batches:
            k = (int(b.y.numel()), -(-int(b.z.numel()) // 8) * 8)
            if k in seen:
                seen[k][0] += 1
            else:
                seen[k] = [1, b]
        return seen

    def precapture(self, seen, keys=None):
        self.captured = sorted(keys)
        self.local = sorted(seen)
        return len(self.captured)


def _worker_precapture(rank, world, port, q, ev):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dig_amd import dp
    from dig_amd.threedgraph.method.run import run
    dp.init_from_env('gloo')
    train = _mols(N_TRAIN, 1)
    n_at = torch.tensor([m.z.numel() for m in train])
    sampler = dp.BalancedBatchSampler(N_TRAIN, BS, rank, world, dp.molecule_cost(n_at), shuffle=True, seed=0)
    r = run()
    r._stepper = _FakeStepper()
    # (1) one rank's scan fails (e.g. check_z_bounds' IndexError on its shard): BOTH ranks raise, nobody hangs in the gather
    class _Broken(_FakeStepper):
        def scan_classes(self, batches):
            if rank == 1:
                raise IndexError('index out of range in self')
            return super().scan_classes(batches)
    r._stepper = _Broken()
    try:
        r._precapture_union(train, sampler, torch.device('cpu'))
        raised = ''
    except RuntimeError as ex:
        raised = str(ex)
    assert 'rank(s) 1: IndexError' in raised, raised
    # (2) the scan is a bounded sample of the plan
    r._stepper = _FakeStepper()
    r.precapture_scan = 2
    assert sum(r._precapture_union(train, sampler, torch.device('cpu'))['local_counts']) <= 2
    del r.precapture_scan
    r._stepper = _FakeStepper()
    rep = r._precapture_union(train, sampler, torch.device('cpu'))
    # the pre-capture pass must not consume the epoch: the plan the trainer iterates afterwards is still epoch 0's
    assert sampler.epoch == 0
    q.put((rank, r._stepper.local, r._stepper.captured, rep))
    ev.wait(60)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_precapture_union_is_the_same_set_on_every_rank():
    """run.py:_precapture_union — every rank scans the size classes of ITS first-epoch batches, the ranks exchange them and
    each is asked to capture the UNION (so that no rank meets a class for the first time at a step where the others wait at
    the all-reduce).  The stepper is a CPU stand-in; the plan, the loader and the object all-gather are the real ones."""
    world = 2
    ctx = mp.get_context('spawn')
    q, ev = ctx.Queue(), ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_worker_precapture, args=(r, world, port, q, ev)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    ev.set()
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, local0, cap0, rep0), (_, local1, cap1, rep1) = res
    assert cap0 == cap1 == sorted(set(map(tuple, local0)) | set(map(tuple, local1)))
    assert rep0['union_classes'] == rep1['union_classes'] == len(cap0) >= max(rep0['local_classes'], rep1['local_classes'])
