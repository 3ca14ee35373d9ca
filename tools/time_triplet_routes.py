"""HIP-event times of the triplet-interaction kernels and of both basis-projection routes (VALU / matrix cores) at the three sizes that
matter: config 2 (32 QM9-like molecules), config 4 (32 OC20-like systems), the roofline launch (512 QM9-like molecules).
Prints one JSON line per (size, kernel)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dig_amd import ops, _hip  # noqa: E402
if os.environ.get('DIG3D_ABL_LIB'):          # an alternative build of the library (same-box A/B of a kernel change)
    from dig_amd import _hip as _h
    _h.LIB_PATH = os.environ['DIG3D_ABL_LIB']
from dig_amd._hip import call, ptr  # noqa: E402
from dig_amd.graph import build_graph, _stream  # noqa: E402
from dig_amd.synthetic import make_batch, batch_to  # noqa: E402
from dig_amd.threedgraph.method.basis import BasisTables  # noqa: E402

PB = 8


def timeit(fn, iters=40, warmup=8):
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
    sizes = {'cfg2_qm9_b32': dict(num_graphs=32, n_min=9, n_max=29, rho=0.08, seed=1),
             'cfg4_oc20_b32': dict(num_graphs=32, n_min=40, n_max=120, rho=0.05, seed=3),
             'roofline_qm9_b512': dict(num_graphs=512, n_min=9, n_max=29, rho=0.08, seed=1)}
    C, ns, nr, nl = 64, 7, 6, 4
    for name, kw in sizes.items():
        b = batch_to(make_batch(cutoff=5.0, **kw), 'cuda')
        g = build_graph(b.pos, b.batch, 5.0, triplets=True)
        N, E, T = g.N, g.E, g.T
        k = g.seg_kj
        gen = torch.Generator().manual_seed(0)
        mk = lambda *sh: torch.randn(*sh, generator=gen).to('cuda')
        X, G, Ps, Pt, w2s, w2t = mk(E, C), mk(E, C), mk(T, PB), mk(T, PB), mk(C, PB), mk(C, PB)
        out, gX, gPs, gPt = torch.empty(E, C, device='cuda'), torch.empty(E, C, device='cuda'), torch.empty(T, PB, device='cuda'), torch.empty(T, PB, device='cuda')
        gW2s, gW2t = torch.empty(C, PB, device='cuda'), torch.empty(C, PB, device='cuda')
        nb_o = max(_hip.query('dig3d_triplet_bwd_blocks', E, C, 0), _hip.query('dig3d_triplet_bwd_blocks', E, C, 1))
        part = torch.empty(nb_o * 2 * C * PB, device='cuda')
        st = _stream()
        runs = {}
        for route, tag in ((1, 'lanegroups'), (2, 'wave'), (3, 'lds'), (0, 'default')):
            runs['trip_fwd_' + tag] = lambda route=route: call('dig3d_triplet_fwd', ptr(X), ptr(g.kj), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(g.tptr), None, E, C, ptr(out), route, st)
            runs['trip_bwdx_' + tag] = lambda route=route: call('dig3d_triplet_fwd', ptr(G), ptr(g.ji), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(k.kptr), ptr(k.perm), E, C, ptr(gX), route, st)
            runs['trip_bwdp_' + tag] = lambda route=route: call('dig3d_triplet_bwd', ptr(G), ptr(X), ptr(g.kj), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(g.tptr), E, C, ptr(gPs), ptr(gPt), ptr(part), ptr(gW2s), ptr(gW2t), 0, route, st)
        # basis projection / weight gradient, both routes
        zeros, norms, pref = BasisTables(ns, nr, 'spherenet').on('cuda')
        posc = b.pos.contiguous()
        dist = ops.edge_dist(posc, g, 0)
        angle, torsion, _ = ops.triplet_geom(posc, g, True)
        bes = ops.bessel_basis(dist, 5.0, ns, nr, zeros, norms, 0)
        KS, KT = ns * nr, ns * ns * nr
        Ws, Wt = mk(KS, 32), mk(KT, 32)
        P1, P2 = torch.empty(nl, T, PB, device='cuda'), torch.empty(nl, T, PB, device='cuda')
        g1, g2 = mk(nl, T, PB), mk(nl, T, PB)
        nbw = _hip.query('dig3d_basis_wgrad_blocks', T)
        partw = torch.empty(nbw * (KS + KT) * 32, device='cuda')
        gWs, gWt = torch.empty(32, KS, device='cuda'), torch.empty(32, KT, device='cuda')
        proj = lambda route: call('dig3d_basis_project', ptr(bes), ptr(g.kj), ptr(angle), ptr(torsion), T, ns, nr, ptr(pref), ptr(Ws), ptr(Wt), nl, ptr(P1), ptr(P2), None, route, st)
        wgr = lambda route: call('dig3d_basis_wgrad', ptr(bes), ptr(g.kj), ptr(angle), ptr(torsion), T, ns, nr, ptr(pref), ptr(g1), ptr(g2), nl, ptr(partw), ptr(gWs), ptr(gWt), None, 0, route, st)
        for route, tag in ((1, 'valu'), (2, 'mfma_4x64'), (0, 'mfma')):
            for nm, fn in (('basis_project_' + tag, proj), ('basis_wgrad_' + tag, wgr)):
                mean, mn = timeit(lambda fn=fn, route=route: fn(route))
                print(json.dumps(dict(size=name, N=N, E=E, T=T, kernel=nm, us_mean=round(mean, 2), us_min=round(mn, 2))), flush=True)
        for nm, fn in runs.items():
            mean, mn = timeit(fn)
            print(json.dumps(dict(size=name, N=N, E=E, T=T, kernel=nm, us_mean=round(mean, 2), us_min=round(mn, 2))), flush=True)


if __name__ == '__main__':
    main()
