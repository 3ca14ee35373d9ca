"""Where does the engine's 30-step trajectory distance (2.1e-5 of the float64 curve on SchNet; the float32 oracle 3.7e-6,
stable under 1-ulp perturbations: tools/diag_trajectory_noise.py) come from?  The same trajectory with one piece swapped
at a time: optimizer (FlatAdam / torch.optim.Adam), execution (HIP-graph replay / eager), and the whole model replaced
by the ORACLE's torch code run on the GPU in float32 (the framework's own GPU kernels: a second reference for what
float32 on this device gives)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dig_amd.synthetic import make_batch, batch_to  # noqa: E402
from dig_amd.graphed import GraphedStep  # noqa: E402
from dig_amd.optim import FlatAdam  # noqa: E402
from tests.fixture_utils import MODEL_CASES, det_state_dict  # noqa: E402
from tests.test_oracle_golden import FWD, oracle_kwargs  # noqa: E402
import dig_amd.threedgraph.method as M  # noqa: E402

torch.set_num_threads(16)
TRAJ = {'schnet_cfg1_b32': dict(n_min=9, n_max=29, cutoff=10.0, batch=32), 'spherenet_tiny': dict(n_min=5, n_max=9, cutoff=5.0, batch=4)}
steps, lr, nb = 30, 5e-4, 6


def oracle_run(case, host, dtype, device='cpu'):
    cls, kw, _, wseed = MODEL_CASES[case]
    model = getattr(M, cls)(**kw)
    sd0 = det_state_dict(model.state_dict(), wseed)
    trainable = {n for n, _ in model.named_parameters()}
    okw = oracle_kwargs(cls, kw)
    sd = {k: (v.clone().to(device=device, dtype=dtype).requires_grad_(k in trainable) if v.is_floating_point() else v.clone().to(device))
          for k, v in sd0.items()}
    opt = torch.optim.Adam([sd[k] for k in sd if k in trainable], lr=lr)
    hb = [batch_to(b, device) for b in host]
    losses = []
    for s in range(steps):
        b = hb[s % nb]
        opt.zero_grad()
        out = FWD[cls](sd, b.z, b.pos, b.batch, dtype=dtype, geom_dtype=torch.float32, **okw)
        loss = (out - b.y.to(dtype).unsqueeze(1)).abs().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return np.array(losses)


def engine_run(case, host, graphed, flat):
    cls, kw, _, wseed = MODEL_CASES[case]
    model = getattr(M, cls)(**kw)
    model.load_state_dict(det_state_dict(model.state_dict(), wseed))
    model = model.to('cuda')
    opt = FlatAdam(model.parameters(), lr=lr) if flat else torch.optim.Adam(model.parameters(), lr=lr)
    dev = [batch_to(b, 'cuda') for b in host]
    stepper = GraphedStep(model) if graphed else None
    losses = []
    for s in range(steps):
        b = dev[s % nb]
        if graphed:
            loss = stepper(b)
        else:
            opt.zero_grad(set_to_none=True)
            loss = (model(b) - b.y.unsqueeze(1)).abs().mean()
            loss.backward()
        opt.step()
        losses.append(loss.item())
    return np.array(losses)


def main():
    for case, t in TRAJ.items():
        host = [make_batch(t['batch'], t['n_min'], t['n_max'], 0.08, t['cutoff'], seed=500 + k) for k in range(nb)]
        l64 = oracle_run(case, host, torch.float64)
        rel = lambda l: float((np.abs(l - l64) / np.abs(l64)).max())
        rec = dict(case=case, oracle32_cpu=rel(oracle_run(case, host, torch.float32)))
        try:
            rec['oracle32_torch_gpu'] = rel(oracle_run(case, host, torch.float32, 'cuda'))
        except Exception as ex:                       # the oracle's shim may be CPU-only
            rec['oracle32_torch_gpu'] = f'{type(ex).__name__}: {str(ex)[:80]}'
        for graphed in (True, False):
            for flat in (True, False):
                rec[f'engine_{"graph" if graphed else "eager"}_{"flatadam" if flat else "torchadam"}'] = rel(engine_run(case, host, graphed, flat))
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
