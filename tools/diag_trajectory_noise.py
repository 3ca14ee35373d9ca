"""How much float32 noise does a 30-step Adam trajectory of the REFERENCE arithmetic carry?  (the engine
sits at 1.8e-5 / 2.1e-5 of the float64 trajectory, the float32 oracle at 5.6e-6 / 3.7e-6 — 'nobody has located where it
comes from'.)  CPU only: the float32 oracle is run from initial weights perturbed by +-1 ulp (relative 6e-8, the size of
ONE rounding error), several seeds; if those runs scatter around the float64 curve as widely as the engine does, the
engine's distance is the amplification of ordinary rounding differences by Adam (update = m / sqrt(v): the direction of a
small-gradient entry flips with its last bits), not an inaccurate kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dig_amd.synthetic import make_batch  # noqa: E402
from tests.fixture_utils import MODEL_CASES, det_state_dict  # noqa: E402
from tests.test_oracle_golden import FWD, oracle_kwargs  # noqa: E402
import dig_amd.threedgraph.method as M  # noqa: E402

torch.set_num_threads(16)
TRAJ = {'spherenet_tiny': dict(n_min=5, n_max=9, cutoff=5.0, batch=4), 'schnet_cfg1_b32': dict(n_min=9, n_max=29, cutoff=10.0, batch=32)}


def run(case, dtype, perturb_seed=None, steps=30, lr=5e-4, nb=6):
    cls, kw, _, wseed = MODEL_CASES[case]
    t = TRAJ[case]
    host = [make_batch(t['batch'], t['n_min'], t['n_max'], 0.08, t['cutoff'], seed=500 + k) for k in range(nb)]
    model = getattr(M, cls)(**kw)
    sd0 = det_state_dict(model.state_dict(), wseed)
    trainable = {n for n, _ in model.named_parameters()}
    okw = oracle_kwargs(cls, kw)
    g = torch.Generator().manual_seed(perturb_seed) if perturb_seed is not None else None
    sd = {}
    for k, v in sd0.items():
        v = v.clone()
        if g is not None and v.is_floating_point() and k in trainable:
            sign = (torch.randint(0, 2, v.shape, generator=g) * 2 - 1).to(v.dtype)
            v = v * (1 + sign * 2.0 ** -24)                       # +-1/2 ulp .. 1 ulp relative
        sd[k] = v.to(dtype).requires_grad_(k in trainable) if v.is_floating_point() else v
    opt = torch.optim.Adam([sd[k] for k in sd if k in trainable], lr=lr)
    losses = []
    for s in range(steps):
        b = host[s % nb]
        opt.zero_grad()
        out = FWD[cls](sd, b.z, b.pos, b.batch, dtype=dtype, geom_dtype=torch.float32, **okw)
        loss = (out - b.y.to(dtype).unsqueeze(1)).abs().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return np.array(losses)


def main():
    for case in TRAJ:
        l64 = run(case, torch.float64)
        l32 = run(case, torch.float32)
        base = float((np.abs(l32 - l64) / np.abs(l64)).max())
        pert = []
        for seed in range(1, 7):
            lp = run(case, torch.float32, perturb_seed=seed)
            pert.append(float((np.abs(lp - l64) / np.abs(l64)).max()))
        print(json.dumps(dict(case=case, oracle32_vs_oracle64=base, perturbed32_vs_oracle64=pert,
                              perturbed_max=max(pert), perturbed_median=float(np.median(pert)))), flush=True)


if __name__ == '__main__':
    main()
