"""GPU box (1 GPU): NCCL(RCCL) process group of world size 1 + watchdog thread alive while HIP graphs are captured
and replayed, with an all-reduce between replay and optimizer step — the DP flow of bench.py / run.py."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29517')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
import dig_amd.threedgraph.method as M
from dig_amd import dp
from dig_amd.graphed import GraphedStep
from dig_amd.synthetic import make_batch, batch_to
torch.cuda.set_device(0)
dist.init_process_group('nccl')
assert dp.is_dist()
torch.manual_seed(0)
m = M.SphereNet(num_layers=2, hidden_channels=64, int_emb_size=32, out_emb_channels=64, num_spherical=3).cuda()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
bucket = dp.GradBucket(m)
st = GraphedStep(m)
bs = [batch_to(make_batch(8, 9, 29, 0.08, 5.0, seed=s), 'cuda') for s in (1, 2, 3)]
t = torch.ones(4, device='cuda'); dist.all_reduce(t)
for i in range(12):
    loss = st(bs[i % 3])
    bucket.allreduce()
    opt.step()
torch.cuda.synchronize()
print('nccl+graph ok, loss', loss.item(), 'captures', st.captures, flush=True)
dist.barrier()
dist.destroy_process_group()
print('done')
