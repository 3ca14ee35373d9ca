"""TEST INFRASTRUCTURE — FULL parameter gradients of the float64-network oracle for cases whose float64 autograd is too
heavy for the GPU box's CPU inside the test run (config 4 at 32 systems: T ~ 6e5 triplets -> [T, 294] float64 tables and
their autograd copies, ~25 GB).  Run once in the build container:

    python -m oracle.make_gradient_golden            # -> tests/golden/grad_<case>.npz

Recorded: energies, the loss, and every parameter's gradient (float64 arithmetic, stored as float32 — 6e-8 relative, two
orders below the 1e-5 the test asks) of loss = mean |out - y| (run.py:127), from the deterministic weights and batch of
tests/fixture_utils.py.  tests/test_gpu_models.py compares the HIP step against it exactly as it compares the other cases
against the oracle evaluated on the spot."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch          # noqa: E402
from tests.test_oracle_golden import oracle_forward                             # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
CASES = ['spherenet_oc20_b32']


def make(case):
    import dig_amd.threedgraph.method as M
    cls, kw, bname, wseed = MODEL_CASES[case]
    assert not kw.get('energy_and_force', False)
    sd = det_state_dict(getattr(M, cls)(**kw).state_dict(), wseed)
    b = get_batch(bname)
    t0 = time.time()
    sd64 = {k: (v.double().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
    out = oracle_forward(cls, sd64, b, torch.float64, torch.float32, kw)
    loss = (out - b.y.double().unsqueeze(1)).abs().mean()
    loss.backward()
    res = {'meta/case': np.asarray(case), 'out': out.detach().numpy(), 'loss': np.asarray(loss.item())}
    for k, v in sd64.items():
        if v.is_floating_point() and v.grad is not None:
            res['grad/' + k] = v.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, 'grad_' + case + '.npz'), **res)
    print(f'{case}: {time.time() - t0:.0f}s loss={loss.item():.6f} params with gradient: {sum(k.startswith("grad/") for k in res)}',
          flush=True)


if __name__ == '__main__':
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    for c in (sys.argv[1:] or CASES):
        make(c)
