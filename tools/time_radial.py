"""HIP-event times of the radial bundle (csrc/radial.hip: dig3d_radial_fwd / _bwd called directly, buffers preallocated) at the
config-2 / config-4 edge counts.  (r06 used it with a `route` argument to compare a column-per-thread backward kernel —
docs/history/r06_radial_bwd2.hip.txt, profiles/r06_radial_bwd_column_form_timing.jsonl: 65.5 vs 53.2 us, not kept.)"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dig_amd import ops, _hip  # noqa: E402
from dig_amd._hip import call, ptr  # noqa: E402
from dig_amd.ops import _ptrs, _stream  # noqa: E402


def timeit(fn, iters=60, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * sum(ms) / len(ms), 1e3 * ms[0]


def main():
    for M in (8704, 36864):
        gen = torch.Generator().manual_seed(0)
        mk = lambda *sh: (torch.randn(*sh, generator=gen) * 0.3).to('cuda')
        K, L, Hc = 6, 4, 128
        x = mk(M, K)
        two = [False, False] + [True, False] * L
        H = len(two)
        Wa = [mk(8, K) if t else mk(Hc, K) for t in two]
        Wb = [mk(Hc, 8) if t else None for t in two]
        bias = [mk(Hc)] + [None] * (H - 1)
        N = [Hc] * H
        J = [8 if t else Hc for t in two]
        act = [1] + [0] * (H - 1)
        IA = ctypes.c_int * H
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        ints = (IA(*N), IA(*J), IA(*act), IA(*[int(t) for t in two]))
        Y = [torch.empty(M, Hc, device='cuda') for _ in range(H)]
        gY = [mk(M, Hc) for _ in range(H)]
        pa, k1 = _ptrs(Wa)
        pb, k2 = _ptrs(Wb)
        pbias, k3 = _ptrs(bias)
        py, k4 = _ptrs(Y)
        pg, k5 = _ptrs(gY)
        st = _stream()
        fwd = lambda: call('dig3d_radial_fwd', ptr(x), M, K, H, pa, pb, pbias, cast(ints[0]), cast(ints[1]), cast(ints[2]), py, st)
        print(json.dumps(dict(M=M, kernel='radial_fwd', us=[round(v, 2) for v in timeit(fwd)])), flush=True)
        stride = _hip.query('dig3d_radial_partial_stride', H, cast(ints[0]), cast(ints[1]), cast(ints[3]), K)
        nb = _hip.query('dig3d_radial_blocks', M, H)
        G = _hip.query('dig3d_radial_bwd_groups', H)
        gX = torch.empty(M, K, device='cuda')
        part = torch.empty(nb * stride, device='cuda')
        work = torch.empty(G * M * K, device='cuda')
        bwd = lambda: call('dig3d_radial_bwd', ptr(x), M, K, H, pa, pb, pbias, cast(ints[0]), cast(ints[1]), cast(ints[2]), pg,
                           ptr(gX), ptr(part), ptr(work), st)
        print(json.dumps(dict(M=M, kernel='radial_bwd (+ gx_sum)', us=[round(v, 2) for v in timeit(bwd)])), flush=True)
        bwd_w = lambda: call('dig3d_radial_bwd', ptr(x), M, K, H, pa, pb, pbias, cast(ints[0]), cast(ints[1]), cast(ints[2]), pg,
                             None, ptr(part), None, st)
        print(json.dumps(dict(M=M, kernel='radial_bwd, weight gradients only (no gX, no gx_sum)',
                              us=[round(v, 2) for v in timeit(bwd_w)])), flush=True)


if __name__ == '__main__':
    main()
