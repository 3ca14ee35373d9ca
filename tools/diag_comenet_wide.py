"""GPU box: ComENet default on a small batch — eager vs GraphedStep losses with the small-M wide route on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

cls, kw, bname, wseed = MODEL_CASES['comenet_default_b8']
print('--- the sequence of test_graphed_step_equals_eager: eager step WITH backward, then the stepper', flush=True)
for wide in (False, True):
    ops.comenet_wide_single = wide
    m = getattr(M, cls)(**kw)
    m.load_state_dict(det_state_dict(m.state_dict(), wseed))
    m = m.cuda()
    b = batch_to(get_batch(bname), 'cuda')
    st = GraphedStep(m)
    st.min_caps = (2 * b.z.numel(), 3 * b.z.numel() * 32 // 2, 20000)
    for it in range(4):
        m.zero_grad()
        loss = (m(b) - b.y.unsqueeze(1)).abs().mean()
        loss.backward()
        ref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        gl = st(b)
        gmax = max(v.abs().max().item() for v in ref.values())
        worst = sorted(((p.grad - ref[n]).abs().max().item() / gmax, n) for n, p in m.named_parameters())[-3:]
        print(f'wide={wide} it={it}: eager {loss.item():.7f} graphed {gl.item():.7f} worst {worst[-1][0]:.2e} {worst[-1][1]}', flush=True)
