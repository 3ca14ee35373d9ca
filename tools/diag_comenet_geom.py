"""GPU box: capture ComENet's geometry step by step on a static graph; which intermediate changes between replays?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd._hip import call, ptr
from dig_amd.graph import start_graph, _stream
from dig_amd.graphed import StaticGraph
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

cls, kw, bname, wseed = MODEL_CASES['comenet_default_b8']
m = getattr(M, cls)(**kw)
m.load_state_dict(det_state_dict(m.state_dict(), wseed))
m = m.cuda()
b = batch_to(get_batch(bname), 'cuda')
g = start_graph(b.pos, b.batch, m.cutoff, triplets=False).finish()
sg = StaticGraph(2 * g.N, 3 * g.N * 16, 0, g.B, b.pos.device, triplets=False)
sg.load(g, b.z, b.pos, b.y)


def steps(sg):
    pos = sg.pos.contiguous()
    st = _stream()
    E, N = sg.E, sg.N
    out = {}
    dist = ops.edge_dist(pos, sg, 1)
    out['dist'] = dist
    add = torch.empty(max(E, 1), dtype=torch.float32, device=pos.device)
    seg = sg.seg_dst
    v0, a0 = ops.segment_argmin(dist, None, seg, E)
    out['v0'], out['a0'] = v0, a0
    call('dig3d_comenet_bump', ptr(a0), N, E, float(m.cutoff), ptr(add), ptr(sg.cnt_N), st)
    out['add_dst'] = add.clone()
    v1, a1 = ops.segment_argmin(dist, add, seg, E)
    out['a1'] = a1
    seg = sg.seg_src
    w0, b0 = ops.segment_argmin(dist, None, seg, E)
    out['b0'] = b0
    call('dig3d_comenet_bump', ptr(b0), N, E, float(m.cutoff), ptr(add), ptr(sg.cnt_N), st)
    out['add_src'] = add.clone()
    w1, b1 = ops.segment_argmin(dist, add, seg, E)
    out['b1'] = b1
    return out


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    steps(sg)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = steps(sg)
gr.replay()
ref = {k: v.clone() for k, v in out.items()}
bad = {k: [] for k in out}
for it in range(30):
    m.zero_grad()
    loss = (m(b) - b.y.unsqueeze(1)).abs().mean()
    loss.backward()
    gr.replay()
    for k, v in out.items():
        if not torch.equal(v, ref[k]):
            nd = int((v != ref[k]).sum())
            bad[k].append((it, nd))
print({k: v[:4] for k, v in bad.items()}, flush=True)
print('live N, E:', g.N, g.E, 'caps', sg.N, sg.E, flush=True)
k = 'a0'
if bad[k]:
    idx = (out[k] != ref[k]).nonzero().flatten()[:10].tolist()
    print('a0 differs at nodes', idx, 'now', out[k][idx].tolist(), 'ref', ref[k][idx].tolist(), flush=True)
k = 'dist'
if bad[k]:
    idx = (out[k] != ref[k]).nonzero().flatten()[:10].tolist()
    print('dist differs at edges', idx, 'now', out[k][idx].tolist(), 'ref', ref[k][idx].tolist(), flush=True)
