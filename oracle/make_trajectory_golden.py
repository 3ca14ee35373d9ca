"""TEST INFRASTRUCTURE — training-TRAJECTORY goldens ("QM9-U0 MAE within 1e-5 of reference").

    python -m oracle.make_trajectory_golden                  # all cases -> tests/golden/traj_<case>.npz
    python -m oracle.make_trajectory_golden spherenet_default_b32

For every case the SAME run is made three times in the build container, from the deterministic weights of
tests/fixture_utils.py over the deterministic batches below, with ``torch.optim.Adam(lr=5e-4)`` and the trainer's loss
(run.py:124-133: L1 on energies, + p * L1 on forces with p = 100 when energy_and_force):

  ref32     the reference's own classes executed VERBATIM (oracle/ref_loader.py) in float32 — the reference's voice;
  oracle32  the restated oracle (oracle/threedgraph_oracle.py) in float32;
  oracle64  the restated oracle with a float64 NETWORK on the float32 geometry — the yardstick both float32 runs and the
            engine are measured against (|ref32 - oracle64| is the float32 noise of such a run).

Recorded: the loss of every step and, after the last step, the energy (and force) MAE of a held-out batch (run.val's
arithmetic, run.py:137-180).  The GPU test (tests/test_gpu_training.py) trains the engine on the same batches and
compares; the GPU box needs neither /root/reference nor minutes of CPU time for the headline model.

Also records ``meta/ref32_s_per_step`` — the verbatim reference's fwd + bwd + Adam wall time per step here (threads in
``meta/threads``): the reference CPU path timed in the build container.
"""
import copy
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fixture_utils import MODEL_CASES, det_state_dict                  # noqa: E402
from tests.trajectory_cases import TRAJ, STEPS, LR, NB, P_FORCE, traj_batches  # noqa: E402
from oracle import ref_loader                                                # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def _verbatim(cls, kw, sd0, host, held, eaf):
    ref_loader.load()
    import digref.threedgraph.method as M
    from torch.autograd import grad
    torch.manual_seed(0)
    model = getattr(M, cls)(**kw)
    model.load_state_dict(sd0)
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    losses, t_steps = [], []
    for s in range(STEPS):
        b = copy.copy(host[s % NB])
        b.pos = b.pos.clone()
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model(b)
        loss = (out - b.y.unsqueeze(1)).abs().mean()
        if eaf:
            force = -grad(outputs=out, inputs=b.pos, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True)[0]
            loss = loss + P_FORCE * (force - b.force).abs().mean()
        loss.backward()
        opt.step()
        t_steps.append(time.perf_counter() - t0)
        losses.append(loss.item())
    b = copy.copy(held)
    b.pos = b.pos.clone()
    model.eval()
    out = model(b)
    f_mae = 0.0
    if eaf:
        force = -grad(outputs=out, inputs=b.pos, grad_outputs=torch.ones_like(out))[0]
        f_mae = (force - b.force).abs().mean().item()
    e_mae = (out.detach() - b.y.unsqueeze(1)).abs().mean().item()
    return np.array(losses), e_mae, f_mae, float(np.median(t_steps[2:]))


def _oracle(cls, kw, sd0, host, held, eaf, dtype, trainable):
    from tests.test_oracle_golden import FWD, oracle_kwargs
    okw = oracle_kwargs(cls, kw)
    sd = {k: (v.clone().to(dtype).requires_grad_(k in trainable) if v.is_floating_point() else v.clone())
          for k, v in sd0.items()}
    params = [sd[k] for k in sd if k in trainable]
    opt = torch.optim.Adam(params, lr=LR)

    def fwd(b, need_force):
        pos = b.pos.clone().requires_grad_(need_force)
        out = FWD[cls](sd, b.z, pos, b.batch, dtype=dtype, geom_dtype=torch.float32, **okw)
        return out, pos

    losses = []
    for s in range(STEPS):
        b = host[s % NB]
        opt.zero_grad()
        out, pos = fwd(b, eaf)
        loss = (out - b.y.to(dtype).unsqueeze(1)).abs().mean()
        if eaf:
            force = -torch.autograd.grad(out, pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
            loss = loss + P_FORCE * (force - b.force.to(force.dtype)).abs().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out, pos = fwd(held, eaf)
    f_mae = 0.0
    if eaf:
        force = -torch.autograd.grad(out, pos, torch.ones_like(out))[0]
        f_mae = (force - held.force.to(force.dtype)).abs().mean().item()
    e_mae = (out.detach() - held.y.to(dtype).unsqueeze(1)).abs().mean().item()
    return np.array(losses), e_mae, f_mae


def make(case):
    warnings.filterwarnings('ignore')
    import dig_amd.threedgraph.method as M
    cls, kw, _, wseed = MODEL_CASES[case]
    eaf = bool(kw.get('energy_and_force', False))
    host, held = traj_batches(case)
    eng = getattr(M, cls)(**kw)                           # (host-side construction only: names, shapes, which are trained)
    sd0 = det_state_dict(eng.state_dict(), wseed)
    trainable = {n for n, _ in eng.named_parameters()}
    t0 = time.time()
    l_ref, e_ref, f_ref, s_per_step = _verbatim(cls, kw, sd0, host, held, eaf)
    l32, e32, f32 = _oracle(cls, kw, sd0, host, held, eaf, torch.float32, trainable)
    l64, e64, f64 = _oracle(cls, kw, sd0, host, held, eaf, torch.float64, trainable)
    # further float32 realisations of the same run: the initial weights moved by <= 1 ulp (a seeded random sign times 2^-24
    # relative).  How far such runs drift from the float64 curve is the float32 NOISE of the run — two realisations (ref32,
    # oracle32) are a thin estimate of it for the chaotic cases (ComENet, the force loss), so three more are recorded
    pert = []
    for k in range(3):
        gen = torch.Generator().manual_seed(7000 + k)
        sdp = {n: (v * (1.0 + (torch.randint(0, 2, v.shape, generator=gen).float() * 2 - 1) * 2.0 ** -24)
                   if (v.is_floating_point() and n in trainable) else v) for n, v in sd0.items()}
        pert.append(_oracle(cls, kw, sdp, host, held, eaf, torch.float32, trainable))
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    out = {'meta/case': np.asarray(case), 'meta/steps': np.asarray(STEPS), 'meta/lr': np.asarray(LR), 'meta/nb': np.asarray(NB),
           'meta/threads': np.asarray(torch.get_num_threads()), 'meta/host_cores': np.asarray(os.cpu_count()),
           'meta/ref32_s_per_step': np.asarray(s_per_step), 'meta/molecules_per_batch': np.asarray(host[0].num_graphs),
           'ref32/loss': l_ref, 'ref32/e_mae': np.asarray(e_ref), 'ref32/f_mae': np.asarray(f_ref),
           'oracle32/loss': l32, 'oracle32/e_mae': np.asarray(e32), 'oracle32/f_mae': np.asarray(f32),
           'oracle64/loss': l64, 'oracle64/e_mae': np.asarray(e64), 'oracle64/f_mae': np.asarray(f64)}
    for k, (lp, ep, fp) in enumerate(pert):
        out.update({f'noise32_{k}/loss': lp, f'noise32_{k}/e_mae': np.asarray(ep), f'noise32_{k}/f_mae': np.asarray(fp)})
    np.savez_compressed(os.path.join(GOLD, 'traj_' + case + '.npz'), **out)
    print(f'{case}: {time.time() - t0:.0f}s  loss {l64[0]:.5f} -> {l64[-1]:.5f}   ref32 vs oracle64 {rel(l_ref, l64):.2e}   '
          f'oracle32 vs oracle64 {rel(l32, l64):.2e}   1-ulp runs vs oracle64 {[float(f"{rel(p[0], l64):.2e}") for p in pert]}   MAE ref32/o32/o64 {e_ref:.6f}/{e32:.6f}/{e64:.6f}   '
          f'verbatim reference {s_per_step * 1e3:.0f} ms/step on {torch.get_num_threads()} threads '
          f'= {host[0].num_graphs / s_per_step:.1f} molecules/s', flush=True)


if __name__ == '__main__':
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    for c in (sys.argv[1:] or list(TRAJ)):
        make(c)
