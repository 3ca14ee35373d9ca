"""HIP-event times of the 256-wide chain (csrc/wide.hip) against the per-layer launches it replaces, at the two shapes the
models run: the output blocks of a default SphereNet forward (G = 5 blocks x 600 atoms) and ComENet's residual layers
(16 384 atoms).  Forward and forward+backward."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dig_amd import ops  # noqa: E402


def timeit(fn, iters=30, warmup=6):
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
    return round(1e3 * sum(ms) / len(ms), 1), round(1e3 * ms[0], 1)


def main():
    gen = torch.Generator().manual_seed(0)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to('cuda')
    for name, G, M, K0, spec in (('readout_5x600', 5, 600, 128, ((0, 0), (1, 0), (1, 0), (1, 0))),
                                 ('comenet_lins_16384', 1, 16384, 256, ((1, 1),) * 4)):
        xs = [mk(M, K0).requires_grad_() for _ in range(G)]
        layers = [[(mk(256, K0 if l == 0 else 256, sc=0.06).requires_grad_(), mk(256, sc=0.1).requires_grad_(),
                    ops.ACT_SWISH if a else ops.ACT_NONE, r) for l, (a, r) in enumerate(spec)] for _ in range(G)]
        cot = [mk(M, 256) for _ in range(G)]
        leaves = xs + [t for ls in layers for (w, b, _, _) in ls for t in (w, b)]

        def wide_f():
            with torch.no_grad():
                return ops.wide_chain(xs, layers)

        def wide_fb():
            with ops.deferred_reductions() as red:
                g = torch.autograd.grad(ops.wide_chain(xs, layers), leaves, cot)
            red.flush()
            return g

        def per_layer(xs_):
            hs = list(xs_)
            for l in range(len(spec)):
                act, res = layers[0][l][2], layers[0][l][3]
                if G > 1:
                    hs = ops._GroupedLinear.apply(act, G, *hs, *[ls[l][0] for ls in layers], *[ls[l][1] for ls in layers])
                else:
                    hs = [ops.linear(hs[0], layers[0][l][0], layers[0][l][1], act, res=hs[0] if res else None)]
            return hs

        def old_f():
            with torch.no_grad():
                return per_layer(xs)

        def old_fb():
            with ops.deferred_reductions() as red:
                g = torch.autograd.grad(per_layer(xs), leaves, cot)
            red.flush()
            return g

        for tag, fn in (('wide_fwd', wide_f), ('layers_fwd', old_f), ('wide_fwd_bwd', wide_fb), ('layers_fwd_bwd', old_fb)):
            mean, mn = timeit(fn)
            print(json.dumps(dict(shape=name, route=tag, us_mean=mean, us_min=mn)), flush=True)


if __name__ == '__main__':
    main()
