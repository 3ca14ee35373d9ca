"""GPU box: which framework-level (ATen) operators are still launched by one eager training step of a bench workload,
with their input shapes and call counts — the map from the `at::native::*` / `Cijk_*` rows of a rocprof kernel table back
to the lines of the model that issue them.   python tools/trace_ops.py [workload]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dig_amd.synthetic import make_batch, batch_to
import dig_amd.threedgraph.method as M

name = sys.argv[1] if len(sys.argv) > 1 else 'dimenetpp_md17_force'
wl = bench.WORKLOADS[name]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
kw = dict(wl['kw'])
if wl['model'] == 'SphereNet':
    kw['num_spherical'] = 7
model = getattr(M, wl['model'])(**kw).to(dev)
b = batch_to(make_batch(wl['batch'], seed=wl['seed'], **wl['gen']), dev)
forces = bool(kw.get('energy_and_force', False))


def step():
    model.zero_grad(set_to_none=True)
    out = model(b)
    loss = (out - b.y.unsqueeze(1)).abs().mean()
    if forces:
        force = -torch.autograd.grad(out, b.pos, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
        loss = loss + 100.0 * (force - b.force).abs().mean()
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
where = {}
for e in prof.events():
    if not e.name.startswith('aten::'):
        continue
    if e.name in ('aten::empty', 'aten::empty_like', 'aten::view', 'aten::as_strided', 'aten::empty_strided', 'aten::reshape',
                  'aten::_unsafe_view', 'aten::unsqueeze', 'aten::select', 'aten::slice', 'aten::t', 'aten::transpose',
                  'aten::expand', 'aten::detach', 'aten::alias', 'aten::contiguous', 'aten::result_type', 'aten::to',
                  'aten::_to_copy', 'aten::resize_', 'aten::squeeze', 'aten::permute', 'aten::narrow', 'aten::item',
                  'aten::_local_scalar_dense', 'aten::view_as', 'aten::unbind', 'aten::numel', 'aten::size', 'aten::stride',
                  'aten::is_nonzero', 'aten::lift_fresh', 'aten::flatten', 'aten::expand_as', 'aten::ones_like',
                  'aten::zeros_like', 'aten::zeros', 'aten::ones', 'aten::new_empty', 'aten::new_zeros', 'aten::matmul',
                  'aten::linear', 'aten::sum_to_size', 'aten::sub', 'aten::rsub'):
        if e.name not in ('aten::sub', 'aten::rsub'):
            continue
    key = (e.name, str(e.input_shapes)[:90])
    cnt[key] += 1
    if key not in where and e.stack:
        fr = [s for s in e.stack if '/root/repo' in s or 'dig_amd' in s]
        where[key] = ' <- '.join(f.split('/')[-1][:60] for f in fr[:3])
tot = sum(cnt.values())
print(f'{name}: {tot} ATen operator calls in one eager step')
for (n, sh), c in cnt.most_common(60):
    print(f'{c:5d}  {n:28s} {sh:90s} {where.get((n, sh), "")}')
