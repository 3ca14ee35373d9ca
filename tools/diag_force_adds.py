"""GPU box: where do the framework's elementwise kernels of the energy_and_force step come from?  One eager step of
DimeNet++ (config 3 shapes) under torch.profiler: aten::add / mul / zeros / fill calls grouped by operand shapes, split
into the forward, the create_graph backward (force) and the final backward."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from dig_amd import ops
from dig_amd.synthetic import batch_to
from tests.fixture_utils import MODEL_CASES, det_state_dict, get_batch
import dig_amd.threedgraph.method as M

case = sys.argv[1] if len(sys.argv) > 1 else 'dimenetpp_force_md17_b32'
cls, kw, bname, wseed = MODEL_CASES[case]
m = getattr(M, cls)(**kw)
m.load_state_dict(det_state_dict(m.state_dict(), wseed))
m = m.cuda()
params = [p for p in m.parameters() if p.requires_grad]


def step(b, mark=False):
    rf = record_function if mark else (lambda n: torch.autograd.profiler.record_function(n))
    with record_function('PHASE_forward'):
        out = m(b)
    with record_function('PHASE_force_backward'):
        f = -torch.autograd.grad(out, b.pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
    with record_function('PHASE_loss'):
        loss = (out - b.y.unsqueeze(1)).abs().mean() + 100 * (f - b.force).abs().mean()
    with record_function('PHASE_final_backward'):
        ops.backward(loss, params)
    return loss


for _ in range(2):
    m.zero_grad(set_to_none=True)
    step(batch_to(get_batch(bname), 'cuda'))
torch.cuda.synchronize()
b = batch_to(get_batch(bname), 'cuda')
m.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step(b)
torch.cuda.synchronize()
evs = [e for e in prof.events()]
phases = sorted([(e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.startswith('PHASE_')])
def phase_of(e):
    for s, t, n in phases:
        if s <= e.time_range.start <= t:
            return n[6:]
    return 'autograd_thread'          # backward nodes run on the engine's thread: attribute by time below
def phase_by_time(e):
    for s, t, n in phases:
        if s <= e.time_range.start <= t:
            return n[6:]
    return '?'
cnt = collections.Counter()
for e in evs:
    if e.name in ('aten::add', 'aten::add_', 'aten::mul', 'aten::zeros', 'aten::zeros_like', 'aten::fill_', 'aten::zero_', 'aten::neg',
                  'aten::sum', 'aten::cat', 'aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::sub', 'aten::div', 'aten::abs', 'aten::sgn'):
        shp = tuple(tuple(s) for s in (e.input_shapes or []) if s)
        cnt[(phase_by_time(e), e.name, shp)] += 1
for (ph, name, shp), c in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f'{ph:16s} {name:18s} x{c:3d}  {shp}')
print('E, T, N =', m and None, flush=True)
