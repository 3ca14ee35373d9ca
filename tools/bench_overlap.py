"""Do two independent low-occupancy kernels of the config-2 step overlap on this part, and in which form?  

Pair A/B = the forward's independent work right after the geometry: A = the basis projection (k_basis_project_mfma<7,true>,
one wave per SIMD, ~57 us); B = the radial projections of the forward (k_radial_fwd) followed by a 128-wide dense layer on the
edge rows (k_linear_fwd<4>) — what the step runs next and what does not depend on A.  Timed five ways, wall clock between
host-visible events over many repetitions (HIP events on the launching stream + a final join):

  serial_eager      A then B on one stream                                                  (what the step does today)
  streams_eager     A on stream 1, B on stream 2, event join, launched kernel by kernel
  graph_serial      the serial sequence captured as ONE HIP graph
  graph_branches    one HIP graph whose capture forked B onto a second stream (parallel branches inside the graph)
  two_graphs        A captured as one graph, B as another, replayed on two streams with an event join

Prints one JSON line with microseconds per repetition for each form and the kernels' stand-alone times."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dig_amd import ops  # noqa: E402
from dig_amd.graph import build_graph  # noqa: E402
from dig_amd.synthetic import make_batch, batch_to  # noqa: E402
import dig_amd.threedgraph.method as M  # noqa: E402


def main(reps=200):
    dev = 'cuda'
    torch.manual_seed(0)
    model = M.SphereNet().to(dev)
    b = batch_to(make_batch(32, 9, 29, 0.08, 5.0, seed=1), dev)
    g = build_graph(b.pos, b.batch, 5.0, triplets=True)
    posc = b.pos.contiguous()
    with torch.no_grad():
        dist, rbf0, bes = model.emb.edge_front(posc, g)
        angle, torsion, _ = ops.triplet_geom(posc, g, True)
        x = torch.randn(g.E, 384, device=dev)
        lin = model.init_e.lin

        def A():
            return model.emb.forward_projected(dist, angle, torsion, g, model.update_es, (rbf0, bes))

        def B():
            rb = model._radial_bundle(rbf0)
            return ops.linear(x, lin.weight, lin.bias, ops.ACT_SWISH), rb

        def timed(fn, n=reps):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n

        res = dict(A_alone_us=timed(A), B_alone_us=timed(B))
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        cur = torch.cuda.current_stream()

        def serial():
            A()
            B()

        def streams():
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                A()
            with torch.cuda.stream(s2):
                B()
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        res['serial_eager_us'] = timed(serial)
        res['streams_eager_us'] = timed(streams)

        def capture(fn):
            gr = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(gr):
                keep = fn()
            return gr, keep

        g_serial, k1 = capture(serial)
        res['graph_serial_us'] = timed(g_serial.replay)

        def branches():
            c = torch.cuda.current_stream()
            s2.wait_stream(c)
            a = A()
            with torch.cuda.stream(s2):
                bb = B()
            c.wait_stream(s2)
            return a, bb

        g_br, k2 = capture(branches)
        res['graph_branches_us'] = timed(g_br.replay)
        g_a, k3 = capture(A)
        g_b, k4 = capture(B)

        def two_graphs():
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                g_a.replay()
            with torch.cuda.stream(s2):
                g_b.replay()
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        res['two_graphs_us'] = timed(two_graphs)
        res['sizes'] = dict(E=g.E, T=g.T)
        p = torch.cuda.get_device_properties(0)
        res['device'] = p.name
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
