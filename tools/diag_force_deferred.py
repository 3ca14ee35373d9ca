"""GPU box: the energy_and_force step of DimeNet++ three ways — loss.backward(), ops.backward (deferred reductions, eager),
GraphedStep — per-parameter gradient differences against the first (which parameter does a route get wrong?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

case = sys.argv[1] if len(sys.argv) > 1 else 'dimenetpp_force_md17_b8'
cls, kw, bname, wseed = MODEL_CASES[case]
m = getattr(M, cls)(**kw)
m.load_state_dict(det_state_dict(m.state_dict(), wseed))
m = m.cuda()
params = [p for p in m.parameters() if p.requires_grad]


def loss_of(b):
    out = m(b)
    f = -torch.autograd.grad(out, b.pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
    return (out - b.y.unsqueeze(1)).abs().mean() + 100 * (f - b.force).abs().mean()


b = batch_to(get_batch(bname), 'cuda')
m.zero_grad(); loss_of(b).backward()
ref = {n: p.grad.clone() for n, p in m.named_parameters()}
gmax = max(v.abs().max().item() for v in ref.values())
b = batch_to(get_batch(bname), 'cuda')
m.zero_grad(set_to_none=True); ops.backward(loss_of(b), params)
worst = sorted(((p.grad - ref[n]).abs().max().item() / gmax, n) for n, p in m.named_parameters())[-4:]
print('ops.backward (deferred, eager) worst:', worst, flush=True)
b = batch_to(get_batch(bname), 'cuda')
st = GraphedStep(m)
st(b)
worst = sorted(((p.grad - ref[n]).abs().max().item() / gmax, n) for n, p in m.named_parameters())[-4:]
print('GraphedStep worst:', worst, 'captures', st.captures, flush=True)
