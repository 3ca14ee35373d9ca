"""GPU box: which outputs go stale when ComENet's replay is interleaved with eager steps (default small-M route)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

cls, kw, bname, wseed = MODEL_CASES['comenet_default_b8']
m = getattr(M, cls)(**kw)
m.load_state_dict(det_state_dict(m.state_dict(), wseed))
m = m.cuda()
b = batch_to(get_batch(bname), 'cuda')
st = GraphedStep(m)
st.min_caps = (2 * b.z.numel(), 3 * b.z.numel() * 32 // 2, 20000)
mode = sys.argv[1] if len(sys.argv) > 1 else 'bwd'
for it in range(10):
    m.zero_grad()
    out = m(b)
    loss = (out - b.y.unsqueeze(1)).abs().mean()
    if mode == 'bwd':
        loss.backward()
    elif mode == 'opsbwd':
        ops.backward(loss, [p for p in m.parameters()])
    gl = st(b)
    e = st.last
    d = (e.out.detach()[:, 0] - out.detach()[:, 0]).abs()
    sg = e.sg
    cks = [int(t.long().sum()) if t.dtype != torch.float32 else round(float(t.double().sum()), 4) for t in (sg.src, sg.dst, sg.rowptr, sg._by_src.kptr, sg._by_src.perm, sg.pos, sg.z, sg.cnt, sg.ptr, sg.batch32, sg.y)]
    wsum = round(sum(float(p.double().sum()) for p in m.parameters()), 6)
    print(f'it={it} captures={st.captures} eager {loss.item():.6f} graphed {gl.item():.6f} max |diff| {d.max().item():.4f} sg {cks} w {wsum}', flush=True)
