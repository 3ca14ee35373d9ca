"""Workload for the PMC passes of the scatter_add roofline (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
the judged kernel k_segsum_sorted<32> at the bench shape, plus a calibration kernel with a KNOWN byte count in the
same access width (float4 streaming copy through k_gather_mul with an identity index: reads 4*M*C + 4*M bytes,
writes 4*M*C bytes) — MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 and
WRITE_SIZE is uncalibrated, so both are scaled by the calibration kernel's known/measured ratio."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
from dig_amd._hip import call, ptr

M, C, seglen = 1 << 22, 128, 17
g = torch.Generator(device='cpu').manual_seed(7)
lens = torch.randint(1, 2 * seglen, (M // seglen + M // (4 * seglen) + 64,), generator=g)
idx = torch.arange(lens.numel()).repeat_interleave(lens)[:M].cuda()
S = int(idx[-1]) + 1
src = torch.randn(M, C, device='cuda')
ident = torch.arange(M, dtype=torch.int32, device='cuda')
out2 = torch.empty(M, C, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    out = ops.scatter(src, idx, dim=0, dim_size=S, assume_sorted=True)
    call('dig3d_gather_mul', ptr(src), ptr(ident), None, None, M, C, ptr(out2), None, st)
torch.cuda.synchronize()
print('M', M, 'C', C, 'S', S, 'alg_bytes', 4 * M * C + 8 * M + 4 * S * C, 'calib_read', 4 * M * C + 4 * M, 'calib_write', 4 * M * C)
