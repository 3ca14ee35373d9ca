"""The layer-chain kernels (csrc/chain.hip) stand-alone at a given row count: the post-aggregation block of an interaction
layer (spherenet.py:172-182) forward + backward on random data, eager launches.  Run it under

    rocprofv3 --kernel-trace --stats -- python tools/bench_chain.py 8704 64        (per-kernel durations)
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python tools/bench_chain.py 8704 64

and it also prints HIP-event times of the forward and of the whole backward (chain_bwd + chain_wgrad + reduction)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dig_amd import _hip  # noqa: E402
if os.environ.get('DIG3D_ABL_LIB'):          # an ablation build of the library (tools/ablate_chain.sh)
    _hip.LIB_PATH = os.environ['DIG3D_ABL_LIB']
from dig_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8704
K0 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev, H = 'cuda', 128
g = torch.Generator().manual_seed(0)
x0 = torch.randn(M, K0, generator=g).to(dev).requires_grad_()
xji = torch.randn(M, H, generator=g).to(dev).requires_grad_()
x1 = torch.randn(M, H, generator=g).to(dev).requires_grad_()
Ws = [(torch.randn(H, K0 if l == 0 else H, generator=g) / (K0 if l == 0 else H) ** 0.5).to(dev).requires_grad_() for l in range(8)]
bs = [None] + [(torch.randn(H, generator=g) * 0.1).to(dev).requires_grad_() for _ in range(7)]
gout = torch.randn(M, H, generator=g).to(dev)
A = ops.ACT_SWISH
kinds = [(1, xji, True), (0, None, False), (2, None, True), (1, x1, True), (0, None, False), (2, None, True),
         (0, None, False), (2, None, True)]
layers = [(Ws[l], bs[l], A, res, t, save) for l, (res, t, save) in enumerate(kinds)]
assert ops.chain_supported(x0, layers)
ev = lambda: torch.cuda.Event(enable_timing=True)
tf = tb = 0.0
for it in range(iters + 5):
    a, b, c = ev(), ev(), ev()
    a.record()
    y = ops.chain(x0, layers)
    b.record()
    y.backward(gout)
    c.record()
    torch.cuda.synchronize()
    if it >= 5:
        tf += a.elapsed_time(b)
        tb += b.elapsed_time(c)
flops = 2.0 * M * (K0 + 7 * H) * H
print(f'M={M} K0={K0}: forward {tf / iters * 1e3:.1f} us ({flops / (tf / iters * 1e-3) / 1e12:.1f} TF), '
      f'backward (dgrad chain + wgrad + reduce, eager) {tb / iters * 1e3:.1f} us')


# the same launches back to back inside HIP graphs (no host gaps): forward alone, and forward + backward
def graph_time(fn, reps=20):
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        fn()
    torch.cuda.current_stream().wait_stream(s_)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    for _ in range(3):
        gr.replay()
    a, b = ev(), ev()
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


def fwd_only():
    with torch.no_grad():
        ops.chain(x0, layers)


def fwd_bwd():
    # fresh leaf aliases: an AccumulateGrad node of an older graph would pull the capture onto another stream
    al = lambda t: None if t is None else t.detach().requires_grad_()
    x0_, xji_, x1_ = al(x0), al(xji), al(x1)
    W_, b_ = [al(w) for w in Ws], [al(b) for b in bs]
    res_t = {id(xji): xji_, id(x1): x1_}
    lay = [(W_[l], b_[l], A, res, res_t.get(id(t)), save) for l, (res, t, save) in enumerate(kinds)]
    y = ops.chain(x0_, lay)
    torch.autograd.grad(y, [x0_, xji_, x1_] + W_ + [b for b in b_ if b is not None], gout)


tf_, tfb = graph_time(fwd_only), graph_time(fwd_bwd)
print(f'GRAPH M={M} K0={K0} lib={os.path.basename(_hip.LIB_PATH)}: forward {tf_:.1f} us, forward+backward {tfb:.1f} us '
      f'(backward incl. wgrad + reduce {tfb - tf_:.1f} us)')
