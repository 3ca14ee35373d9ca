"""Diagnostic (GPU box): where do HIP geometry values differ from the CPU oracle, and by how much."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pyg_shim as S, threedgraph_oracle as O
from tests.fixture_utils import get_batch
from dig_amd.threedgraph.utils import xyz_to_dat

print('cpu capability', torch.backends.cpu.get_cpu_capability(), 'threads', torch.get_num_threads())
b = get_batch('qm9_b32')
ei = S.radius_graph(b.pos, 5.0, b.batch)
ref = O.xyz_to_dat(b.pos, ei, b.pos.size(0), True)
got = [t.cpu() for t in xyz_to_dat(b.pos.cuda(), ei.cuda(), b.pos.size(0), use_torsion=True)]
j, i = ei
P = b.pos.numpy()
d = P[i] - P[j]
seq = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
def ulps(a, b_):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64); b_ = np.asarray(b_, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b_)
for name, x in (('hip', got[0].numpy()), ('oracle', ref[0].numpy())):
    u = ulps(x, seq)
    print(f'dist {name} vs numpy-seq: mismatches {int((u>0).sum())}/{u.size} max ulp {int(u.max())}')
u = ulps(got[0].numpy(), ref[0].numpy()); print('dist hip vs oracle mismatches', int((u > 0).sum()), 'max ulp', int(u.max()))
# torch-CPU sub-steps
dd = (b.pos[i] - b.pos[j])
print('sub equal', np.array_equal(dd.numpy(), d))
sq = dd.pow(2); print('pow equal', np.array_equal(sq.numpy(), d * d))
sm = sq.sum(-1); print('sum seq equal', np.array_equal(sm.numpy(), (d[:,0]*d[:,0]+d[:,1]*d[:,1])+d[:,2]*d[:,2]))
print('sqrt equal', np.array_equal(sm.sqrt().numpy(), np.sqrt(sm.numpy())))
a = (got[1] - ref[1]).abs(); print('angle max abs diff', a.max().item(), 'n>2e-6', int((a > 2e-6).sum()))
t = (got[2] - ref[2]).abs(); print('torsion max abs diff', t.max().item(), 'n>1e-5', int((t > 1e-5).sum()), 'of', t.numel())
bad = (t > 1e-5).nonzero().view(-1)[:10]
print('bad torsion samples hip/oracle', got[2][bad].tolist(), ref[2][bad].tolist())
