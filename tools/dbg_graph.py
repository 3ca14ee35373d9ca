import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dig_amd.threedgraph.method as M
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import make_batch, batch_to
order = sys.argv[1].split(',')
keep = []
b = batch_to(make_batch(4, 5, 9, 0.08, 5.0, seed=11), 'cuda')
for name in order:
    torch.manual_seed(0)
    kw = dict(hidden_channels=32, int_emb_size=16, out_emb_channels=32, num_spherical=3, num_radial=4, num_layers=2)
    bs = int(os.environ.get('BS', '8'))
    if name == 'DimeNetPP':
        kw['basis_emb_size'] = bs
    else:
        kw.update(basis_emb_size_dist=bs, basis_emb_size_angle=bs, basis_emb_size_torsion=bs)
    m = getattr(M, name)(**kw).cuda()
    st = GraphedStep(m)
    if os.environ.get('CAPS'):
        st.min_caps = tuple(int(v) for v in os.environ['CAPS'].split(','))
    e1 = os.environ.get('EAGER1', '')
    pre = os.environ.get('PRE', '')
    hold = []
    if 'G' in pre:
        from dig_amd.graph import build_graph
        hold.append(build_graph(b.pos, b.batch, 5.0))
    if 'L' in pre:
        hold.append(torch.nn.functional.linear(torch.randn(100, 256, device='cuda'), torch.randn(1, 256, device='cuda')))
    if 'M' in pre:
        from dig_amd import ops
        hold.append(ops.linear(torch.randn(100, 128, device='cuda'), torch.randn(128, 128, device='cuda'), None, 1))
    if 'K' in pre:   # a big allocation kept alive
        hold.append(torch.randn(1 << 24, device='cuda'))
    if 'Z' in pre:   # forward under no_grad
        with torch.no_grad():
            hold.append(m(b))
    if e1:
        if 'U' in e1: m.fused_triplets = False
        out = m(b)
        if 'F' not in e1: (out - b.y.unsqueeze(1)).abs().mean().backward()
        m.fused_triplets = True
        if 'S' in e1:
            import gc
            del out
            for p in m.parameters(): p.grad = None
            torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    for i in range(3):
        l = st(b)
    torch.cuda.synchronize()
    print(name, 'ok loss', l.item(), 'captures', st.captures, flush=True)
    if len(sys.argv) > 2:
        keep.append(st)
    else:
        del st, m
        import gc; gc.collect(); torch.cuda.empty_cache()
print('done')
