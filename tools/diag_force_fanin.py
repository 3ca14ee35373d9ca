"""GPU box: WHICH tensors of the energy_and_force step receive more than one gradient (= one framework addition each, per
pass)?  Builds the step's two autograd graphs (the forward's, walked from `out`; the final one, walked from `loss`) and
counts, for every (node, input slot), how many edges of the graph point at it: a slot with n > 1 incoming edges costs n - 1
`at::native` additions when the pass runs.  Prints the producing node's name, the slot and the tensor shape."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops, diffops
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

case = sys.argv[1] if len(sys.argv) > 1 else 'dimenetpp_force_md17_b32'
cls, kw, bname, wseed = MODEL_CASES[case]
m = getattr(M, cls)(**kw)
m.load_state_dict(det_state_dict(m.state_dict(), wseed))
m = m.cuda()
b = batch_to(get_batch(bname), 'cuda')


def fan_in(root, title):
    seen, stack = set(), [root]
    cnt = collections.Counter()
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        for nxt, slot in fn.next_functions:
            if nxt is None:
                continue
            cnt[(nxt, slot)] += 1
            stack.append(nxt)
    rows = collections.Counter()
    for (fn, slot), c in cnt.items():
        if c > 1 and 'AccumulateGrad' not in fn.name():
            meta = getattr(fn, '_input_metadata', None)
            shp = ''
            try:
                shp = tuple(fn._input_metadata[slot].shape)
            except Exception:
                pass
            rows[(fn.name(), slot, shp, c - 1)] += 1
    print(f'== {title}: {sum(k[3] * v for k, v in rows.items())} additions over {len(seen)} nodes')
    for (name, slot, shp, adds), n in sorted(rows.items(), key=lambda kv: (-kv[0][3] * kv[1], kv[0][0])):
        print(f'  {n:3d} x {name:36s} output {slot:2d} {str(shp):16s} +{adds} each')


out = m(b)
fan_in(out.grad_fn, 'force gradient pass (graph of out)')
with diffops.force_gradient_scope():
    gpos = torch.autograd.grad(out, b.pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
loss = ops.ef_l1_loss(out, b.y.unsqueeze(1), gpos, b.force, None, 100.0) if hasattr(ops, 'ef_l1_loss') else None
fan_in(loss.grad_fn, 'final backward (graph of loss)')
