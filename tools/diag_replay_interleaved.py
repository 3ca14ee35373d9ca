"""GPU box: eager step (forward + backward) interleaved with GraphedStep replays of the SAME batch, 8 rounds — does a
replay ever differ from the eager step?  (found: ComENet's replay went stale after 2-3 rounds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

cases = sys.argv[1:] or ['comenet_default_b8:comenet_group_rows=0', 'comenet_default_b8', 'spherenet_tiny', 'schnet_cfg1_b32', 'comenet_dense128']
for spec in cases:
    case, *flags = spec.split(':')
    saved = {}
    for f in flags:
        k, v = f.split('=')
        saved[k] = getattr(ops, k)
        setattr(ops, k, int(v))
    cls, kw, bname, wseed = MODEL_CASES[case]
    m = getattr(M, cls)(**kw)
    m.load_state_dict(det_state_dict(m.state_dict(), wseed))
    m = m.cuda()
    b = batch_to(get_batch(bname), 'cuda')
    st = GraphedStep(m)
    st.min_caps = (2 * b.z.numel(), 3 * b.z.numel() * 32 // 2 if cls == 'ComENet' else 2000, 20000)
    bad = []
    for it in range(int(os.environ.get('ROUNDS', '8'))):
        m.zero_grad()
        loss = (m(b) - b.y.unsqueeze(1)).abs().mean()
        loss.backward()
        gl = st(b)
        if abs(gl.item() - loss.item()) > 1e-6 * max(1.0, abs(loss.item())):
            bad.append((it, round(gl.item(), 6)))
    print(f'{spec}: eager {loss.item():.6f} replays differing: {bad}', flush=True)
    for k, v in saved.items():
        setattr(ops, k, v)
