"""GPU: what a TRAINING RUN needs beyond single-step parity:
  * trajectory parity — N Adam steps of the engine (HIP-graph replay + FlatAdam) against the CPU oracle restatement
    trained with torch.optim.Adam from identical weights on identical batches: loss curve and held-out MAE
    (north_star "QM9-U0 MAE within 1e-5 of reference": with no dataset in the image this is the evaluable form,
    SURVEY §8c; run.py:103-135);
  * the device loader (run.py:53-55,123 replacement): slot recycling, every tensor attribute travels, validation over
    more batches than the loader has slots."""
import numpy as np
import pytest
import torch

from tests.fixture_utils import MODEL_CASES, det_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _molecules(n, seed, n_min=5, n_max=9, cutoff=5.0, with_force=False):
    from types import SimpleNamespace
    from dig_amd.synthetic import make_batch
    big = make_batch(n, n_min, n_max, 0.08, cutoff, seed=seed, with_force=with_force)
    data = []
    for g in range(n):
        a, b = int(big.ptr[g]), int(big.ptr[g + 1])
        s = SimpleNamespace(z=big.z[a:b], pos=big.pos[a:b], y=big.y[g:g + 1])
        if with_force:
            s.force = big.force[a:b]
        data.append(s)
    return data


from tests.trajectory_cases import TRAJ, STEPS, LR, NB, P_FORCE, traj_batches   # noqa: E402


@pytest.mark.parametrize('case', list(TRAJ))
def test_training_trajectory_matches_reference(case):
    """30 Adam steps (lr 5e-4, run.py:47 defaults) over 6 rotating batches, the engine stepping exactly as run.train steps
    it (HIP-graph replay -> FlatAdam for SchNet / DimeNet++ / SphereNet, the double backward of energy_and_force inside
    the replay; ComENet: ops.backward -> FlatAdam), against tests/golden/traj_<case>.npz (oracle/make_trajectory_golden.py,
    build container):
      ref32     the reference's own classes run VERBATIM in float32 with torch.optim.Adam,
      oracle32  the restated oracle in float32,
      oracle64  the restated oracle with a float64 network on the float32 geometry (the yardstick),
      noise32_k three more float32 runs of the oracle from initial weights moved by <= 1 ulp;
    then the MAE of a held-out batch (run.val's arithmetic).  The largest |run - oracle64| over the five float32
    realisations of the same algorithm is the float32 noise of such a run: 3.7e-6 ... 8.2e-6 of the loss for the two small
    cases.  The engine is held to max(1e-5, 1.5 x that floor) of the float64 curve at EVERY step and on the
    held-out MAE — north_star's "MAE within 1e-5 of reference" in the form that can be evaluated without a dataset.
    Cases: the two of r04 plus the models the metric is quoted on — SphereNet at its defaults, B = 32 (BASELINE config 2 as
    benchmarked), DimeNet++ with forces through the energy_and_force loss (config 3's model), ComENet at its defaults."""
    from dig_amd.synthetic import batch_to
    from dig_amd.graphed import GraphedStep
    from dig_amd.optim import FlatAdam
    from dig_amd import ops
    import dig_amd.threedgraph.method as M
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'traj_' + case + '.npz'))
    assert int(gold['meta/steps']) == STEPS and int(gold['meta/nb']) == NB and float(gold['meta/lr']) == LR
    cls, kw, _, wseed = MODEL_CASES[case]
    eaf = bool(kw.get('energy_and_force', False))
    host, held = traj_batches(case)
    model = getattr(M, cls)(**kw)
    model.load_state_dict(det_state_dict(model.state_dict(), wseed))
    model = model.to(DEV)
    opt = FlatAdam(model.parameters(), lr=LR)
    graphable = cls in ('DimeNetPP', 'SphereNet', 'SchNet', 'ComENet')          # run.py:run — the same rule
    stepper = GraphedStep(model, p=P_FORCE) if graphable else None
    params = [q for q in model.parameters() if q.requires_grad]
    dev = [batch_to(b, DEV) for b in host]
    le = []
    for s in range(STEPS):
        if stepper is not None:
            loss = stepper(dev[s % NB], prefetch=dev[(s + 1) % NB])
        else:                                                        # run.train's kernel-by-kernel branch
            opt.zero_grad()
            b = dev[s % NB]
            loss = (model(b) - b.y.unsqueeze(1)).abs().mean()
            ops.backward(loss, params)
        opt.step()
        le.append(loss.item())
    le = np.array(le)
    model.eval()
    hb = batch_to(held, DEV)
    if eaf:
        hb.pos.requires_grad_(True)
        out = model(hb)
        force = -torch.autograd.grad(out, hb.pos, torch.ones_like(out))[0]
        f_mae = (force - hb.force).abs().mean().item()
    else:
        with torch.no_grad():
            out = model(hb)
        f_mae = 0.0
    e_mae = (out.detach() - hb.y.unsqueeze(1)).abs().mean().item()
    if stepper is not None:
        assert stepper.captures <= 6 and not stepper.disabled       # one graph per size class of the six batches
    l64, lref, l32 = gold['oracle64/loss'], gold['ref32/loss'], gold['oracle32/loss']
    relv = lambda a, b: np.abs(a - b) / np.abs(b)
    relmax = lambda a, b: float(relv(a, b).max())
    # the float32 noise of the run UP TO step s (running maximum over the two float32 realisations): early steps are held to
    # 1e-5 even where the late steps of a case are chaotic (Adam's first updates are lr * sign(g): a parameter whose true
    # gradient is ~0 moves by +-lr on rounding noise; L1 force residuals change sign) — ComENet and the force case reach
    # 1e-3 between the reference's own float32 run and the float64 curve by step 30, SphereNet stays at 1.8e-6
    f32_runs = [lref, l32] + [gold[k] for k in sorted(gold.files) if k.startswith('noise32_') and k.endswith('/loss')]
    ref_floor_s = np.maximum.accumulate(relv(lref, l64))                 # the VERBATIM reference's own float32 noise, running
    all_floor_s = np.maximum.accumulate(np.max([relv(r, l64) for r in f32_runs], axis=0))
    # The yardstick is the verbatim reference's float32 run: 1.5 x its running distance from the float64 curve wherever that
    # run stays in the linear regime (<= 1e-4 at step 30: SphereNet 1.8e-6, SchNet 3.8e-6, and ALSO the force case, 1.5e-5 —
    # r05 took the maximum over five realisations there, one of which, the RESTATED oracle in float32, wanders to 2.8e-3: a
    # tolerance of 8e-3 that a 1e-3 regression would have passed).  Only ComENet is chaotic in the reference's OWN arithmetic
    # (1.1e-3 by step 30: Adam's first updates are lr * sign(g), a parameter whose true gradient is ~0 moves by +-lr on
    # rounding noise); there the floor is the maximum over the five float32 realisations and the factor 3 — they differ
    # from EACH OTHER by up to 6x at single steps, and four of the five share one implementation (torch CPU kernels, weights
    # moved by 1 ulp), which samples input rounding but not summation order: an independent implementation lands up to ~2x
    # outside their maximum (measured: 1.9x at step 6).
    chaotic = float(ref_floor_s[-1]) > 1e-4
    floor_s = all_floor_s if chaotic else ref_floor_s
    floor = float(floor_s[-1])
    rel_s = relv(le, l64)
    rel = float(rel_s.max())
    factor = 3.0 if chaotic else 1.5
    tol_s = np.maximum(1e-5, factor * floor_s)
    m64 = float(gold['oracle64/e_mae']) + P_FORCE * float(gold['oracle64/f_mae'])
    mref = float(gold['ref32/e_mae']) + P_FORCE * float(gold['ref32/f_mae'])
    m32 = float(gold['oracle32/e_mae']) + P_FORCE * float(gold['oracle32/f_mae'])
    me = e_mae + P_FORCE * f_mae
    mks = sorted({k.split('/')[0] for k in gold.files if k.startswith('noise32_')})
    m_noise = [float(gold[k + '/e_mae']) + P_FORCE * float(gold[k + '/f_mae']) for k in mks]
    mae_floor = (max(abs(v - m64) for v in [mref, m32] + m_noise) if chaotic else abs(mref - m64)) / abs(m64)
    rep = dict(loss_first=le[0], loss_last=le[-1], loss_rel_vs_oracle64=rel, loss_rel_vs_reference32=relmax(le, lref),
               reference32_vs_oracle64=relmax(lref, l64), oracle32_vs_oracle64=relmax(l32, l64),
               mae_engine=me, mae_reference32=mref, mae_oracle64=m64, mae_rel_vs_oracle64=abs(me - m64) / abs(m64),
               mae_rel_vs_reference32=abs(me - mref) / abs(mref), mae_floor=mae_floor,
               captures=stepper.captures if stepper is not None else 0,
               steps_held_to_1e5=int((tol_s <= 1e-5).sum()), worst_ratio_to_tolerance=float((rel_s / tol_s).max()),
               floor_factor=factor, floor_from_verbatim_reference_only=0.0 if chaotic else 1.0,
               tolerance_last_step=float(tol_s[-1]))
    from tests.test_gpu_models import _report
    _report('trajectory_' + case, **rep)
    assert le[-NB:].mean() < le[:NB].mean(), rep        # it trains: the last pass over the batches against the first
    # whatever the late steps do, step 0 (identical weights, no update yet) is single-step parity: 1e-5 in every case (from
    # step 1 on the verbatim reference's own float32 run is already 1.4e-5 / 1.7e-5 off the float64 curve in the force and
    # ComENet cases: Adam's first update is lr * sign(g))
    assert float(rel_s[0]) <= 1e-5, (rep, float(rel_s[0]))
    assert bool((rel_s <= tol_s).all()), (rep, 'first step outside the tolerance:', int(np.argmax(rel_s > tol_s)),
                                          rel_s.tolist(), tol_s.tolist())
    assert rep['mae_rel_vs_oracle64'] <= max(1e-5, factor * mae_floor), rep


def test_device_loader_recycles_slots_without_corrupting_live_batches():
    """DeviceLoader (dig_amd/threedgraph/data.py): 11 batches through 4 slots, the consumer holding the current batch
    while it already has the next one (run.train's look-ahead, lag = 1).  Every batch must read back exactly its own
    bytes at the time it is used, custom tensor attributes travel like ``batch.to(device)`` would move them, and
    ``ptr_list`` stays a host list."""
    from dig_amd.threedgraph.data import DataLoader, DeviceLoader
    data = _molecules(44, seed=31, with_force=True)
    for i, s in enumerate(data):
        s.node_feature = torch.full((s.z.numel(), 2), float(i))
    host = list(DataLoader(data, 4, shuffle=False))
    assert len(host) == 11
    dl = DeviceLoader(DataLoader(data, 4, shuffle=False), torch.device(DEV), depth=4, lag=1)
    it = iter(dl)
    cur = next(it)
    k = 0
    burn = torch.randn(2048, 2048, device=DEV)
    while cur is not None:
        nxt = next(it, None)
        (burn @ burn).sum()                              # keep the compute stream busy between uses
        for key in ('z', 'pos', 'batch', 'y', 'force', 'node_feature', 'ptr'):
            assert torch.equal(getattr(cur, key).cpu(), getattr(host[k], key)), (k, key)
        assert cur.ptr_list == host[k].ptr_list and cur.num_graphs == 4
        cur, k = nxt, k + 1
    assert k == 11
    # a second epoch over the same loader object re-uses the slots
    assert sum(1 for _ in dl) == 11


def test_validation_over_more_batches_than_loader_slots():
    """run.val keeps predictions AND targets of every batch until the final MAE (run.py:171-180).  The targets are views
    of recycled loader slots: with more validation batches than slots they must be copies."""
    import dig_amd.threedgraph.method as M
    from dig_amd.threedgraph.evaluation import ThreeDEvaluator
    from dig_amd.threedgraph.data import DataLoader
    for eaf in (False, True):
        data = _molecules(36, seed=33, with_force=eaf)
        torch.manual_seed(0)
        model = M.SchNet(num_layers=2, hidden_channels=32, num_filters=32, cutoff=5.0, energy_and_force=eaf).to(DEV)
        r = M.run()
        got = r.val(model, DataLoader(data, 3, shuffle=False), eaf, 100, ThreeDEvaluator(), torch.device(DEV))
        # the plain route: one batch at a time, .to(device), nothing shared
        e_err = f_err = 0.0
        n_at = 0
        for b in DataLoader(data, 3, shuffle=False):
            b = b.to(DEV)
            out = model(b)
            if eaf:
                f = -torch.autograd.grad(out, b.pos, torch.ones_like(out))[0]
                f_err += (f - b.force).abs().sum().item()
                n_at += b.force.numel()
            e_err += (out.detach() - b.y.unsqueeze(1)).abs().sum().item()
        want = e_err / len(data) + (100 * f_err / n_at if eaf else 0.0)
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (eaf, got, want)
