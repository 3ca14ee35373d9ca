"""GPU box: capture STAGES of ComENet's forward on a static graph as separate HIP graphs, interleave eager full steps, and
report which stage's replay output ever changes (a dangling read inside the captured region)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd.graph import start_graph
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


def geom(sg):
    return torch.stack(m.geometry(sg.pos.contiguous(), sg))


def feats(sg):
    d, th, ph, ta = m.geometry(sg.pos.contiguous(), sg)
    f1, f2 = m.features(d, th, ph, ta)
    return f1.sum(1) + f2.sum(1)


def emb(sg):
    return m.emb(sg.z.long())


def block0(sg):
    d, th, ph, ta = m.geometry(sg.pos.contiguous(), sg)
    f1, f2 = m.features(d, th, ph, ta)
    x = m.emb(sg.z.long())
    return m.interaction_blocks[0](x, f1, f2, sg)


def conv_only(sg):
    d, th, ph, ta = m.geometry(sg.pos.contiguous(), sg)
    f1, f2 = m.features(d, th, ph, ta)
    x = m.emb(sg.z.long())
    blk = m.interaction_blocks[0]
    x = blk.lin(x, blk.act)
    wc = blk.lin_feature1.composed_weight(f1)
    agg = ops.feature_conv(x, f1, wc, sg.seg_src, sg.seg_dst)
    return agg


def full(sg):
    return m(sg)


stages = dict(geom=geom, feats=feats, emb=emb, conv_only=conv_only, block0=block0, full=full)
graphs = {}
with torch.no_grad():
    for name, fn in stages.items():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(sg)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn(sg)
        gr.replay()
        graphs[name] = (gr, out, out.clone())
bad = {k: [] for k in stages}
for it in range(int(os.environ.get('ROUNDS', '30'))):
    m.zero_grad()
    loss = (m(b) - b.y.unsqueeze(1)).abs().mean()
    loss.backward()
    for name, (gr, out, ref) in graphs.items():
        gr.replay()
        live = slice(0, None)
        d = (out - ref)
        d = d[torch.isfinite(ref)]
        if d.numel() and d.abs().max().item() > 0:
            bad[name].append(it)
print({k: v[:6] for k, v in bad.items()}, flush=True)
