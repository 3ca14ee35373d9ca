"""First library call after which a tensor it was handed holds inf / NaN — run under the fence allocator
(DIG3D_EFENCE=hi|lo python tools/efence/trace_nan.py): the payload of every allocation is 0x7f7f7f7f (finite, 3.39e38), so
a non-finite value means a kernel consumed bytes it should not have (slack past a logical end, an unwritten slot).
Every C-ABI call is followed by a device synchronisation; the tensors whose pointers were passed are checked before and
after.  TEST INFRASTRUCTURE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if os.environ.get('DIG3D_EFENCE'):
    so = os.path.join(ROOT, 'tools', 'efence', 'libefence.so')
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(so, 'efence_malloc', 'efence_free'))

from dig_amd import _hip  # noqa: E402
import dig_amd.ops, dig_amd.graph, dig_amd.diffops, dig_amd.graphed, dig_amd.optim  # noqa: E402,E401

registry = {}
orig_ptr, orig_call = _hip.ptr, _hip.call
hits = [0]


def ptr2(t):
    if t is not None and torch.is_tensor(t) and t.is_cuda and t.is_floating_point() and t.numel():
        registry[t.data_ptr()] = t
    return orig_ptr(t)


def finite(t):
    return bool(torch.isfinite(t).all())


def call2(name, *args):
    mine = [(i, registry[a]) for i, a in enumerate(args) if isinstance(a, int) and a in registry]
    torch.cuda.synchronize()
    before = {i: finite(t) for i, t in mine}
    orig_call(name, *args)
    torch.cuda.synchronize()
    bad = [(i, t) for i, t in mine if not finite(t)]
    if bad and hits[0] < 6:
        hits[0] += 1
        print(f'[trace] {name}: non-finite after the call in args '
              + ', '.join(f'#{i}{tuple(t.shape)}(finite before: {before[i]}, bad elems {int((~torch.isfinite(t)).sum())}, '
                          f'first bad flat index {int((~torch.isfinite(t)).flatten().nonzero()[0])})' for i, t in bad), flush=True)
        print('        all tensor args: ' + ', '.join(f'#{i}{tuple(t.shape)}' for i, t in mine), flush=True)


for mod in (dig_amd.ops, dig_amd.graph, dig_amd.diffops, dig_amd.graphed, dig_amd.optim, _hip):
    if hasattr(mod, 'ptr'):
        mod.ptr = ptr2
    if hasattr(mod, 'call'):
        mod.call = call2

from dig_amd.synthetic import make_batch, batch_to  # noqa: E402
import dig_amd.threedgraph.method as M  # noqa: E402

torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'SphereNet'
kw = dict(hidden_channels=64, int_emb_size=32, out_emb_channels=64, num_spherical=3, num_radial=4, num_layers=2)
if which == 'SchNet':
    kw = dict(num_layers=2, hidden_channels=32, num_filters=32)
if which == 'ComENet':
    kw = dict(num_layers=2, hidden_channels=64, middle_channels=32)
model = getattr(M, which)(**kw).to('cuda:0')
b = batch_to(make_batch(4, 6, 12, 0.08, 5.0, seed=5), 'cuda:0')
out = model(b)
torch.cuda.synchronize()
print('[trace] forward out', out.flatten().tolist(), flush=True)
(out - b.y.unsqueeze(1)).abs().mean().backward()
torch.cuda.synchronize()
bad = [n for n, p in model.named_parameters() if p.grad is not None and not finite(p.grad)]
print('[trace] non-finite parameter gradients:', bad, flush=True)
