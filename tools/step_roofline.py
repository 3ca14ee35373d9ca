"""Algorithmic flops and bytes of ONE training step (forward + loss + backward + Adam) of a bench workload, counted from
the reference's own operator list — what `bench.py` divides by the measured step time to give `step_roofline`
(north_star: throughput "as absolute molecules/s and as fraction of the HBM roofline"; SURVEY.md §8d "Algorithmic flops").

The count is per REFERENCE OPERATOR (method/spherenet/spherenet.py, dimenetpp/dimenetpp.py, comenet/comenet.py,
schnet/schnet.py), not per kernel of this engine:

  dense layer on R rows, K -> N      flops 2 R K N,     bytes 4 (R K + R N + K N)        (input, output, weight, once each)
  fused triplet interaction          flops 35 T C (torsion) / 18 T C,  bytes SURVEY §8d's fused figure
                                     4 E C + 4 T (b_a + b_t) + 16 T + 4 E C              (x_kj, projected bases, idx_kj/idx_ji int64, out)
  basis rows (never materialised)    flops of the first basis Linears 2 T (ns nr b_a + ns^2 nr b_t),  bytes 4 T (2 angles + 1 index) + 4 E ns nr
                                     + their [T, b] outputs (the [T, ns^2 nr] rows the reference writes are NOT counted: they
                                     need not exist)
  ComENet lin_feature1/2             two bias-free Linears without activation = one [H, f] map per edge: 2 E f H flops on the
                                     vector units inside the convolution, 4 E f bytes (not the [E, H] tensor the reference writes)
  scatter_add / segment sums         flops M C,         bytes 4 M C + 8 M + 4 S C        
  elementwise products / gathers     flops R C,         bytes 4 * 3 R C

and per pass: the backward of a dense layer is two products of the forward's size (input gradient, weight gradient), the
backward of a gather / scatter / product moves the forward's bytes once more per operand => backward = 2 x forward for
flops AND bytes; the energy_and_force step (run.py:124-133: forward, create_graph backward w.r.t. positions, then the
backward of both) is counted as forward x (1 + 1 + 2 + 2) = 6 x dense forward flops (the position-gradient pass has no
weight products; its backward and the final backward have both).  Adam: 4 reads + 3 writes of P floats, 10 P flops.

`mfma_frac` = dense flops / time / 157.3 TF (float32 matrix peak = vector peak, MI355X_MICROARCH.md); `hbm_frac` = bytes /
time / 8 TB/s.  A fused engine moves FEWER bytes than this operator-level figure (the chain kernels keep eight layers'
activations on chip), so hbm_frac is "how fast the reference's operator traffic is retired", not measured traffic.
"""

F32_TFLOPS = 157.3
HBM_GBS = 8000.0


class _Acc:
    def __init__(self):
        self.dense = 0.0      # flops of dense layers (matrix-core work)
        self.other = 0.0      # flops of everything else (VALU work)
        self.bytes = 0.0
        self.items = {}

    def lin(self, tag, R, K, N, count=1, input_on_chip=False):
        f, b = 2.0 * R * K * N * count, 4.0 * ((0 if input_on_chip else R * K) + R * N + K * N) * count
        self.dense += f
        self.bytes += b
        d = self.items.setdefault(tag, [0.0, 0.0])
        d[0] += f
        d[1] += b

    def op(self, tag, flops, bytes_):
        self.other += flops
        self.bytes += bytes_
        d = self.items.setdefault(tag, [0.0, 0.0])
        d[0] += flops
        d[1] += bytes_


def _dime_forward(kw, N, E, T, B, torsion):
    H = kw.get('hidden_channels', 128)
    L = kw.get('num_layers', 4)
    ie = kw.get('int_emb_size', 64)
    bd = kw.get('basis_emb_size_dist', kw.get('basis_emb_size', 8))
    ba = kw.get('basis_emb_size_angle', kw.get('basis_emb_size', 8))
    bt = kw.get('basis_emb_size_torsion', kw.get('basis_emb_size', 8)) if torsion else 0
    oe = kw.get('out_emb_channels', 256)
    ns, nr = kw.get('num_spherical', 7), kw.get('num_radial', 6)
    nb, na = kw.get('num_before_skip', 1), kw.get('num_after_skip', 2)
    no = kw.get('num_output_layers', 3)
    oc = kw.get('out_channels', 1)
    a = _Acc()
    # geometry + basis (geometric_computing.py:12-80, features.py): reads pos, writes dist/angle/torsion; basis rows are
    # consumed by the first basis Linears without touching HBM
    a.op('geometry', 30.0 * T + 10.0 * E, 4.0 * (3 * N + E + (2 if torsion else 1) * T) + 8.0 * (2 * E + 2 * T))
    a.op('dist_emb+bessel', 20.0 * E * (nr + ns * nr), 4.0 * E * (1 + nr + ns * nr))
    # init (spherenet.py:53-91)
    a.op('embedding gather', 0.0, 4.0 * (2 * E * H))
    a.lin('init.lin_rbf_0', E, nr, H)
    a.lin('init.lin', E, 3 * H, H)
    a.lin('init.lin_rbf_1', E, nr, H)
    a.op('e2 = rbf1 * e1', E * H, 12.0 * E * H)
    for _ in range(L):
        # update_e (spherenet.py:94-182)
        a.lin('lin_ji, lin_kj', E, H, H, 2)
        a.lin('lin_rbf1, lin_rbf2', E, nr, bd)
        a.lin('lin_rbf1, lin_rbf2', E, bd, H)
        a.op('x_kj * rbf', E * H, 12.0 * E * H)
        a.lin('lin_down', E, H, ie)
        a.lin('lin_sbf1 / lin_t1 (basis rows in registers)', T, ns * nr, ba, input_on_chip=True)
        if torsion:
            a.lin('lin_sbf1 / lin_t1 (basis rows in registers)', T, ns * ns * nr, bt, input_on_chip=True)
        a.op('basis evaluation', T * (ns * nr + (ns * ns * nr if torsion else 0)) * 4.0, 4.0 * T * (3 if torsion else 2) + 4.0 * E * ns * nr)
        a.op('triplet interaction (lin_sbf2, lin_t2, gather, products, scatter)', (35.0 if torsion else 18.0) * T * ie,
             4.0 * E * ie + 4.0 * T * (ba + bt) + 16.0 * T + 4.0 * E * ie)
        a.lin('lin_up', E, ie, H)
        a.lin('residual layers + lin', E, H, H, 2 * nb + 1 + 2 * na)
        a.lin('lin_rbf', E, nr, H)
        a.op('e2 = rbf * e1', E * H, 12.0 * E * H)
    for _ in range(L + 1):
        # update_v / update_u (spherenet.py:185-225)
        a.op('edge -> node scatter_add', E * H, 4.0 * E * H + 8.0 * E + 4.0 * N * H)
        a.lin('output blocks', N, H, oe)
        a.lin('output blocks', N, oe, oe, no)
        a.lin('output blocks', N, oe, oc)
        a.op('node -> graph scatter_add', N * oc, 4.0 * N * oc + 8.0 * N + 4.0 * B * oc)
    return a


def _comenet_forward(kw, N, E, B):
    H = kw.get('hidden_channels', 256)
    L = kw.get('num_layers', 4)
    mid = kw.get('middle_channels', 64)
    ns, nr = kw.get('num_spherical', 2), kw.get('num_radial', 3)
    no = kw.get('num_output_layers', 3)
    oc = kw.get('out_channels', 1)
    f1, f2 = nr * ns * ns, nr * ns
    a = _Acc()
    a.op('geometry (4 arg-min searches, theta / phi / tau) + bases', 80.0 * E, 4.0 * (3 * N + 5 * E) + 8.0 * 2 * E + 4.0 * E * (f1 + f2))
    a.op('embedding', 0.0, 4.0 * N * H)
    for _ in range(L):
        # SimpleInteractionBlock (comenet.py:136-215)
        # lin_feature1 / lin_feature2 are two bias-free Linears with no activation between them (comenet.py:87-112,
        # act=False): algorithmically ONE [H, f] map per edge, 2 E f H flops, whose [E, H] result need not exist in HBM — the
        # convolution consumes it (4 E f bytes of features instead of 4 E H of weights).  Counted with the convolution.
        a.op('EdgeGraphConv: w_e = Wc f_e, gather * w_e -> scatter (conv1 + conv2)', 2.0 * E * H * (f1 + f2) + 2.0 * 2 * E * H,
             4.0 * E * (f1 + f2) + 2 * (4.0 * E + 4.0 * N * H + 4.0 * N * H))
        a.lin('conv lin_rel / lin_root', N, H, H, 4)
        a.lin('block dense layers', N, H, H, 1 + 2)          # lin, lin1, lin2
        a.lin('block dense layers', N, 2 * H, H)             # lin_cat
        a.lin('block dense layers', N, H, H, 3)              # the three residual lins
        a.lin('block dense layers', N, H, H)                 # final
        a.op('GraphNorm', 8.0 * N * H, 4.0 * 2 * N * H)
    a.lin('output', N, H, H, no)
    a.lin('output', N, H, oc)
    a.op('node -> graph scatter_add', N * oc, 4.0 * N * oc + 8.0 * N + 4.0 * B * oc)
    return a


def _schnet_forward(kw, N, E, B):
    H = kw.get('hidden_channels', 128)
    F = kw.get('num_filters', 128)
    L = kw.get('num_layers', 6)
    G = kw.get('num_gaussians', 50)
    a = _Acc()
    a.op('distances + gaussians', 10.0 * E * G, 4.0 * (3 * N + E + E * G) + 16.0 * E)
    a.op('embedding', 0.0, 4.0 * N * H)
    for _ in range(L):
        a.lin('filter network', E, G, F)
        a.lin('filter network', E, F, F)
        a.lin('cfconv lin', N, H, F)
        a.op('cfconv gather * W -> scatter_add', 2.0 * E * F, 4.0 * E * F + 4.0 * E * F + 8.0 * E + 4.0 * N * F)
        a.lin('update_v', N, F, H)
        a.lin('update_v', N, H, H)
    a.lin('readout', N, H, H // 2)
    a.lin('readout', N, H // 2, 1)
    a.op('node -> graph scatter_add', N, 4.0 * N + 8.0 * N + 4.0 * B)
    return a


def step_roofline(model_name, kw, sizes, n_params, ms_per_step, forces=False):
    """-> dict for the bench line.  ``sizes`` = dict(N, E, T, B) (mean over the batches the timed loop cycles)."""
    N, E, T, B = (float(sizes[k]) for k in ('N', 'E', 'T', 'B'))
    if model_name in ('SphereNet', 'DimeNetPP'):
        a = _dime_forward(kw, N, E, T, B, torsion=model_name == 'SphereNet')
    elif model_name == 'ComENet':
        a = _comenet_forward(kw, N, E, B)
    elif model_name == 'SchNet':
        a = _schnet_forward(kw, N, E, B)
    else:
        return dict(error=f'no operator list for {model_name}')
    # passes: forward 1, backward 2 (input + weight gradients); with forces: forward 1, position-gradient pass 1, the
    # backward of both 2 + 2
    mult = 6.0 if forces else 3.0
    adam_f, adam_b = 10.0 * n_params, 4.0 * 7 * n_params
    dense, other, byt = a.dense * mult, a.other * mult + adam_f, a.bytes * mult + adam_b
    t = ms_per_step * 1e-3
    top = sorted(a.items.items(), key=lambda kv: -kv[1][0])[:6]
    return dict(
        flops=dense + other, flops_dense=dense, flops_other=other, bytes=byt, passes=mult,
        achieved_tflops=(dense + other) / t / 1e12, achieved_dense_tflops=dense / t / 1e12, achieved_gbs=byt / t / 1e9,
        mfma_peak_tflops=F32_TFLOPS, hbm_peak_gbs=HBM_GBS,
        mfma_frac=dense / t / 1e12 / F32_TFLOPS, valu_frac=other / t / 1e12 / F32_TFLOPS, hbm_frac=byt / t / 1e9 / HBM_GBS,
        sizes=dict(N=N, E=E, T=T, B=B, parameters=int(n_params)),
        largest_forward_terms={k: dict(gflop=v[0] / 1e9, mbytes=v[1] / 1e6) for k, v in top},
        formula='per reference operator, once per pass: dense 2RKN flops / 4(RK+RN+KN) bytes; fused triplet op 35TC (18TC without '
                'torsion) / 4EC+4T(ba+bt)+16T+4EC; scatter_add MC / 4MC+8M+4SC; products RC / 12RC; x3 passes (forward, input '
                'gradients, weight gradients; x6 with forces) + Adam 10P / 28P — tools/step_roofline.py, DESIGN.md §6')
