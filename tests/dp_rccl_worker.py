"""Launched by tests/test_gpu_dp_rccl.py under ``python -m torch.distributed.run`` — one process per GPU over RCCL.

Checks on real ranks what tests/test_dp_gloo.py checks with a stand-in on CPU: the data-parallel gradient
of an ENGINE model — produced by the graphed step into its flat buffer, pre-scaled by B_local / B_global through
``set_scale`` (unequal shards), summed by ONE RCCL all-reduce — equals the single-GPU gradient of the mean loss over the
concatenated batch, and FlatAdam then moves every replica to bit-identical weights.  Exit code 0 = pass."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def take_graphs(b, g0, g1):
    """graphs [g0, g1) of a host batch as a batch of their own."""
    ptr = b.ptr_list
    a, e = ptr[g0], ptr[g1]
    out = SimpleNamespace(z=b.z[a:e], pos=b.pos[a:e], batch=b.batch[a:e] - g0, y=b.y[g0:g1], num_graphs=g1 - g0,
                          node_feature=None, ptr_list=[p - a for p in ptr[g0:g1 + 1]])
    if hasattr(b, 'force'):
        out.force = b.force[a:e]
    return out


def main():
    from dig_amd import dp
    from dig_amd.graphed import GraphedStep
    from dig_amd.optim import FlatAdam
    from dig_amd.synthetic import make_batch, batch_to
    from tests.fixture_utils import det_state_dict
    import dig_amd.threedgraph.method as M
    # DPW_ONE_GPU=1: every rank on device 0, the collective over gloo (RCCL refuses two ranks on one device) — the ENGINE
    # (HIP kernels, GraphedStep, union pre-capture, the after_replay hook, FlatAdam) under a real world of 2 on a 1-GPU box;
    # only the transport of the one flat buffer differs from the 8-GPU job
    one_gpu = os.environ.get('DPW_ONE_GPU') == '1'
    rank, world = dp.init_from_env('gloo' if one_gpu else 'nccl')
    assert dp.is_dist(), 'launch under torch.distributed.run (or DIG3D_FORCE_DIST=1 for a 1-rank group)'
    dev = torch.device('cuda', 0 if one_gpu else int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    tiny = dict(n_min=5, n_max=9, rho=0.08, per=4)
    cases = [('SchNet', dict(num_layers=2, hidden_channels=32, num_filters=32, cutoff=5.0), 3e-6, tiny),
             ('SphereNet', dict(hidden_channels=32, int_emb_size=16, out_emb_channels=32, num_spherical=3, num_radial=4,
                                num_layers=2, basis_emb_size_dist=4, basis_emb_size_angle=4, basis_emb_size_torsion=4), 3e-6, tiny)]
    if world > 1:
        # BASELINE config-2- and config-4-shaped shards: the default SphereNet (hidden 128, num_spherical 7) on QM9-like
        # molecules, 16 (+1) per rank, and on OC20-like 40-120-atom systems, 4 (+1) per rank
        cases += [('SphereNet', dict(), 1e-5, dict(n_min=9, n_max=29, rho=0.08, per=16)),
                  ('SphereNet', dict(), 1e-5, dict(n_min=40, n_max=120, rho=0.05, per=4))]
    for cls, kw, tol, gen in cases:
        per = gen['per']
        B = per * world + (1 if world > 1 else 0)       # unequal shards: rank 0 holds one graph more
        host = make_batch(B, gen['n_min'], gen['n_max'], gen['rho'], 5.0, seed=41)
        cut = [0] + [per * (r + 1) + (1 if world > 1 else 0) for r in range(world)]
        mine = take_graphs(host, cut[rank], cut[rank + 1])
        torch.manual_seed(1000 + rank)                    # replicas are BUILT apart; the weights below make them equal
        model = getattr(M, cls)(**kw)
        model.load_state_dict(det_state_dict(model.state_dict(), 7))
        model = model.to(dev)
        # single-GPU reference: eager step on the whole batch
        full = batch_to(host, dev)
        model.zero_grad(set_to_none=True)
        loss = (model(full) - full.y.unsqueeze(1)).abs().mean()
        loss.backward()
        ref = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        # data parallel: graphed step on the local shard, flat buffer pre-scaled by B_local / B_global, one all-reduce
        opt = FlatAdam(model.parameters(), lr=1e-3)
        dp.broadcast_parameters(model, opt)
        bucket = dp.GradBucket(model)
        stepper = GraphedStep(model, grad_scale=1.0 / world)
        stepper.set_scale(mine.num_graphs / B)
        local = batch_to(mine, dev)
        for it in range(2):                               # capture, then a pure replay
            if it == 0 or os.environ.get('DPW_SYNC'):
                stepper(local)
                bucket.allreduce_flat(stepper.flat)
            else:                                         # the training loop's form: asynchronous, started behind the replay
                stepper(local, after_replay=bucket.allreduce_flat_start)
                bucket.allreduce_flat_finish()
            gmax = max(v.abs().max().item() for v in ref.values())
            worst = max((p.grad - ref[n]).abs().max().item() for n, p in model.named_parameters()) / gmax
            assert worst <= tol, (cls, it, worst)
        assert not stepper.disabled and stepper.captures == 1
        opt.step()
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        w0 = w.clone()
        dist.broadcast(w0, 0)
        assert torch.equal(w, w0), f'{cls}: replicas diverged after one step'
        # exact global MAE from ragged shards (run.val's reduction)
        s = dp.allreduce_scalar_sum(float(mine.num_graphs), dev)
        assert s == float(B)
        if rank == 0:
            print(f'dp_rccl_worker: {cls} world={world} B={B} atoms={gen["n_min"]}-{gen["n_max"]} hidden={kw.get("hidden_channels", 128)} '
                  f'worst grad err {worst:.2e} OK', flush=True)
    run_api_leg(rank, world, dev)
    run_api_leg(rank, world, dev, oc20_shaped=True)
    dist.barrier()
    dist.destroy_process_group()


def run_api_leg(rank, world, dev, oc20_shaped=False):
    """(``oc20_shaped``: BASELINE config-4-shaped shards — the default SphereNet, hidden 128, on 40-120-atom systems, whose
    size classes really differ from batch to batch and from rank to rank: the union pre-capture under RCCL is exercised
    with several classes, and NO capture may happen after it during the epochs.)

    SphereNet through the TRAINER's data-parallel branch (run.run: balanced plan, ragged last global batch weighted by
    B_local / B_global, size classes of the whole job captured before step 0, async all-reduce behind the replay):
    every replica must end on bit-identical weights, with no capture after the pre-capture pass in epoch 1."""
    from dig_amd.synthetic import make_batch
    from dig_amd.threedgraph.method.run import run
    from dig_amd.threedgraph.evaluation import ThreeDEvaluator
    from tests.fixture_utils import det_state_dict
    import dig_amd.threedgraph.method as M

    def mols(n, seed):
        b = make_batch(n, 40, 120, 0.05, 5.0, seed=seed) if oc20_shaped else make_batch(n, 5, 9, 0.08, 5.0, seed=seed)
        p = b.ptr_list
        return [SimpleNamespace(z=b.z[p[i]:p[i + 1]], pos=b.pos[p[i]:p[i + 1]], y=b.y[i:i + 1]) for i in range(n)]

    bs = 4
    n_train = 2 * bs * world + (bs * world - 1)           # two full global batches + a ragged one (one graph short)
    torch.manual_seed(500 + rank)
    model = (M.SphereNet() if oc20_shaped else
             M.SphereNet(hidden_channels=32, int_emb_size=16, out_emb_channels=32, num_spherical=3, num_radial=4,
                         num_layers=2, basis_emb_size_dist=4, basis_emb_size_angle=4, basis_emb_size_torsion=4))
    if rank == 0:
        model.load_state_dict(det_state_dict(model.state_dict(), 9))     # the other ranks must RECEIVE these
    r = run()
    r.run(dev, mols(n_train, 61), mols(5, 62), mols(5, 63), model, torch.nn.L1Loss(), ThreeDEvaluator(), epochs=2,
          batch_size=bs, vt_batch_size=3, lr=1e-3)
    assert r._stepper is not None and not r._stepper.disabled
    rep = getattr(r, 'precapture_report', None)
    if world > 1:
        assert rep is not None and rep['captured'] >= 1 and rep['union_classes'] >= rep['local_classes'], rep
        if oc20_shaped:
            # the deterministic plan of epoch 0 was scanned whole (3 batches per rank): its classes were all captured before
            # step 0; epoch 1 reshuffles and may meet new ones, which are captured on first sight — bounded by the batches
            assert rep['captured'] <= r._stepper.captures <= rep['captured'] + 3, (rep, r._stepper.captures)
    w = torch.cat([q.detach().reshape(-1) for q in model.parameters()])
    w0 = w.clone()
    dist.broadcast(w0, 0)
    assert torch.equal(w, w0), 'run.run: replicas diverged'
    assert torch.isfinite(w).all() and r.best_valid == r.best_valid
    if rank == 0:
        print(f'dp_rccl_worker: run.run{" (config-4-shaped)" if oc20_shaped else ""} world={world} captures={r._stepper.captures} precapture={rep} OK', flush=True)


if __name__ == '__main__':
    main()
