"""ComENet on the HIP engine.  Drop-in for ``dig.threedgraph.method.ComENet``
(method/comenet/comenet.py:218-402): same constructor keywords/defaults, ``forward(batch_data)``, and
``state_dict`` layout.

HIP: radius graph, the four scatter_min reference-neighbour searches (comenet.py:304-327), theta/phi/tau
(:329-385), the Bessel x harmonics features (comenet/features.py), and every EdgeGraphConv
``sum_j edge_weight * x_j`` (:130-133) as a fused gather-multiply-segment-sum, GraphNorm as one kernel per pass
(csrc/norm.hip), dense Linears + bias + swish (+ residual) on the f32-MFMA kernels (csrc/dense.hip).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ..._hip import call, ptr
from ...graph import build_graph, _stream
from .basis import BasisTables
from ..data import check_z_bounds
from .inits import glorot_


def swish(x):
    return F.silu(x)


class Linear(nn.Module):
    """comenet.py:29-84 — glorot weight, zero bias by default."""

    def __init__(self, in_channels, out_channels, bias=True, weight_initializer='glorot',
                 bias_initializer='zeros'):
        super().__init__()
        assert in_channels > 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer, self.bias_initializer = weight_initializer, bias_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        wi = self.weight_initializer
        if wi == 'glorot':
            glorot_(self.weight)
        elif wi == 'glorot_orthogonal':
            from .inits import glorot_orthogonal_
            glorot_orthogonal_(self.weight, 2.0)
        elif wi == 'uniform':
            bound = 1.0 / math.sqrt(self.weight.size(-1))
            nn.init.uniform_(self.weight.data, -bound, bound)
        elif wi in ('kaiming_uniform', None):
            from .inits import kaiming_uniform_
            kaiming_uniform_(self.weight, self.in_channels, math.sqrt(5))
        elif wi == 'zeros':
            self.weight.data.zero_()
        else:
            raise RuntimeError(f"Linear layer weight initializer '{wi}' is not supported")
        if self.bias is not None:
            if self.bias_initializer == 'zeros':
                self.bias.data.zero_()
            elif self.bias_initializer is None:
                bound = 1.0 / math.sqrt(self.in_channels)
                self.bias.data.uniform_(-bound, bound)
            else:
                raise RuntimeError(f"Linear layer bias initializer '{self.bias_initializer}' is not supported")

    def forward(self, x, act=None, res=None):
        """act(x W^T + b) (+ res) — one f32-MFMA kernel (csrc/dense.hip); act: None or swish."""
        return ops.linear(x, self.weight, self.bias, ops.ACT_SWISH if act is swish else ops.ACT_NONE, res)


class TwoLayerLinear(nn.Module):
    def __init__(self, in_channels, middle_channels, out_channels, bias=False, act=False):
        super().__init__()
        self.lin1 = Linear(in_channels, middle_channels, bias=bias)
        self.lin2 = Linear(middle_channels, out_channels, bias=bias)
        self.act = act

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    compose = True          # False: the two layers one after the other (tests compare the routes)

    def composable(self, x):
        """the two layers collapse into one small-K layer: no bias, no activation, in <= 16"""
        return (self.compose and not self.act and self.lin1.bias is None and self.lin2.bias is None and x.is_cuda
                and x.dim() == 2 and self.lin1.weight.size(1) <= 16 and self.lin2.weight.size(0) <= 256
                and self.lin2.weight.size(0) % 8 == 0)

    def composed_weight(self, x):
        """W2 W1 [out, in] when ``composable``, else None."""
        if self.composable(x):
            return ops.compose_weights([(self.lin2.weight, self.lin1.weight)])[0]
        return None

    def forward(self, x):
        if self.composable(x):
            # two bias-free Linears with nothing between them (comenet.py:50-52, act=False): applied as ONE layer with
            # W2 W1 on the small-K kernel — the [E, middle] intermediate and the E-row middle -> hidden GEMM (E = 5e5 rows
            # at 128 atoms x 128 molecules: 17 GFLOP per call, 8 calls per step) are never formed; the factor gradients
            # follow from the [hidden, K] product by autograd
            return ops.linear(x, self.composed_weight(x))
        x = self.lin1(x, swish if self.act else None)
        return self.lin2(x, swish if self.act else None)


class EmbeddingBlock(nn.Module):
    def __init__(self, hidden_channels, act=swish):
        super().__init__()
        self.act = act
        self.emb = nn.Embedding(95, hidden_channels)
        self.reset_parameters()

    def reset_parameters(self):
        self.emb.weight.data.uniform_(-math.sqrt(3), math.sqrt(3))

    def forward(self, x):
        return self.act(ops.embedding(x, self.emb.weight))


class EdgeGraphConv(nn.Module):
    """PyG GraphConv(aggr='add') with message = edge_weight * x_j (comenet.py:130-133):
    lin_rel(sum_{j->i} w_e * x_j) + lin_root(x_i).  PyG Linear init = kaiming_uniform(a=sqrt 5)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    fused_features = True   # False: the [E, hidden] edge-weight tensor route (tests compare the routes)

    def forward(self, x, g, feature, lin_feature, wc=None):
        """``lin_feature(feature)`` is the edge weight of the reference (comenet.py:171-172); ``wc``: its composed weight
        when the model formed the composed weights of all blocks in one launch."""
        if wc is None and self.fused_features:
            wc = lin_feature.composed_weight(feature)
        # x has three consumers per convolution (the aggregation, lin_root, and whatever comes after this module): each
        # hands an alias of x on to the next, so the three gradients are summed inside the backward kernels
        if wc is not None and ops.feature_conv_supported(x, feature, wc):
            # the edge weight Wc f_e is evaluated inside the aggregation kernel: no [E, hidden] tensor in either pass
            agg, x = ops.feature_conv(x, feature, wc, g.seg_src, g.seg_dst, tap=True)
        else:
            agg = ops.gather_mul_segment_sum(x, lin_feature(feature), None, g.seg_src, g.seg_dst)
        root, x = ops.linear_tap(x, self.lin_root.weight)
        return ops.linear(agg, self.lin_rel.weight, self.lin_rel.bias, ops.ACT_NONE, res=root), x


class GraphNorm(nn.Module):
    """PyG GraphNorm as the fused per-graph kernel pair of csrc/norm.hip."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def reset_parameters(self):
        self.weight.data.fill_(1)
        self.bias.data.zero_()
        self.mean_scale.data.fill_(1)

    def forward(self, x, g):
        # one workgroup per graph: mean, variance and the affine output in one launch; backward in one more
        return ops.graph_norm(x, self.weight, self.bias, self.mean_scale, g.ptr, g.B, self.eps, padded=g.cnt_N is not None)


def _wide(xs, chains):
    """G independent chains of 256-wide layers — ``chains[g]`` = [(layer, act), ...], the same spec for every group — through
    the 256-wide chain kernel (csrc/wide.hip): at a few hundred atoms a [600, 256] x [256, 256] layer costs ~7 us there
    against ~20 us on the tiled dense kernel (64 x 128 tiles: 20 blocks on 256 CUs).  -> list of the G outputs, or None
    when the shapes do not fit (the caller keeps its per-layer route)."""
    layers = [[(lin.weight, lin.bias, ops.ACT_SWISH if a is swish else ops.ACT_NONE, 0) for lin, a in ch] for ch in chains]
    if ops._wide_chain and ops.comenet_wide_single and ops.wide_chain_supported(list(xs), layers):
        return list(ops.wide_chain(list(xs), layers))
    return None


class SimpleInteractionBlock(nn.Module):
    """comenet.py:136-215."""

    def __init__(self, hidden_channels, middle_channels, num_radial, num_spherical, num_layers, output_channels,
                 act=swish):
        super().__init__()
        self.act = act
        self.conv1 = EdgeGraphConv(hidden_channels, hidden_channels)
        self.conv2 = EdgeGraphConv(hidden_channels, hidden_channels)
        self.lin1 = Linear(hidden_channels, hidden_channels)
        self.lin2 = Linear(hidden_channels, hidden_channels)
        self.lin_cat = Linear(2 * hidden_channels, hidden_channels)
        self.norm = GraphNorm(hidden_channels)
        self.lin_feature1 = TwoLayerLinear(num_radial * num_spherical ** 2, middle_channels, hidden_channels)
        self.lin_feature2 = TwoLayerLinear(num_radial * num_spherical, middle_channels, hidden_channels)
        self.lin = Linear(hidden_channels, hidden_channels)
        self.lins = nn.ModuleList([Linear(hidden_channels, hidden_channels) for _ in range(num_layers)])
        self.final = Linear(hidden_channels, output_channels)
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.conv1, self.conv2, self.norm, self.lin_feature1, self.lin_feature2, self.lin, self.lin1,
                  self.lin2, self.lin_cat, *self.lins, self.final):
            m.reset_parameters()

    def forward(self, x, feature1, feature2, g, wc=(None, None)):
        small = x.size(0) < ops.comenet_group_rows
        y = _wide([x], [[(self.lin, self.act)]]) if small else None
        x = y[0] if y is not None else self.lin(x, self.act)
        c1 = self.conv1
        if (x.size(0) < ops.comenet_group_rows and self.act is swish and wc[0] is not None and wc[1] is not None
                and ops.feature_conv_supported(x, feature1, wc[0]) and ops.feature_conv_supported(x, feature2, wc[1])
                and ops.grouped_linear_supported([x, x], [c1.lin_root.weight, self.conv2.lin_root.weight])
                and self.lin1.weight.shape == self.lin2.weight.shape and c1.lin_rel.weight.shape == c1.lin_root.weight.shape):
            # a few hundred atoms (the reference's QM9 runs): every launch is ~20 us of latency whatever its size, so the
            # block's three pairs of independent same-shape layers run as three grouped launches per pass instead of six
            c2 = self.conv2
            agg1, x = ops.feature_conv(x, feature1, wc[0], g.seg_src, g.seg_dst, tap=True)
            agg2, x = ops.feature_conv(x, feature2, wc[1], g.seg_src, g.seg_dst, tap=True)
            grouped = bool(ops.comenet_group_pairs)      # (False: the per-layer launches, for same-box comparisons)
            y = _wide([x, x], [[(c1.lin_root, None)], [(c2.lin_root, None)]]) if grouped else None
            if y is not None:
                root1, root2 = y
            elif grouped:
                root1, root2 = ops.grouped_linear([x, x], [c1.lin_root.weight, c2.lin_root.weight], [None, None])
            else:
                root1, x = ops.linear_tap(x, c1.lin_root.weight)
                root2, x = ops.linear_tap(x, c2.lin_root.weight)
            if grouped:
                c1o, c2o = ops.grouped_linear([agg1, agg2], [c1.lin_rel.weight, c2.lin_rel.weight],
                                              [c1.lin_rel.bias, c2.lin_rel.bias], ops.ACT_NONE, [root1, root2])
            else:
                c1o = ops.linear(agg1, c1.lin_rel.weight, c1.lin_rel.bias, ops.ACT_NONE, res=root1)
                c2o = ops.linear(agg2, c2.lin_rel.weight, c2.lin_rel.bias, ops.ACT_NONE, res=root2)
            y = _wide([c1o, c2o], [[(self.lin1, self.act)], [(self.lin2, self.act)]]) if grouped else None
            if y is not None:
                h1, h2 = y
            elif grouped:
                h1, h2 = ops.grouped_linear([c1o, c2o], [self.lin1.weight, self.lin2.weight], [self.lin1.bias, self.lin2.bias],
                                            ops.ACT_SWISH)
            else:
                h1, h2 = self.lin1(c1o, self.act), self.lin2(c2o, self.act)
        else:
            c1, x = self.conv1(x, g, feature1, self.lin_feature1, wc[0])      # (convolution, alias of x for the next consumer)
            c2, x = self.conv2(x, g, feature2, self.lin_feature2, wc[1])
            h1 = self.lin1(c1, self.act)
            h2 = self.lin2(c2, self.act)
        h = ops.linear_cat2(h1, h2, self.lin_cat.weight, self.lin_cat.bias, res=x)   # lin_cat(cat([h1, h2], 1)) + x
        layers = [[(lin.weight, lin.bias, ops.ACT_SWISH if self.act is swish else ops.ACT_NONE, 1) for lin in self.lins]]
        if ops._wide_chain and layers[0] and ops.wide_chain_supported([h], layers):
            (h,) = ops.wide_chain([h], layers)       # the residual layers as one launch per pass (csrc/wide.hip)
        else:
            for lin in self.lins:
                h = lin(h, self.act, res=h)
        h = self.norm(h, g)
        y = _wide([h], [[(self.final, None)]]) if small else None
        return y[0] if y is not None else self.final(h)


class ComENet(nn.Module):
    r"""ComENet (`"Towards Complete and Efficient Message Passing for 3D Molecular Graphs"`); API of
    method/comenet/comenet.py:232-242."""

    def __init__(self, cutoff=8.0, num_layers=4, hidden_channels=256, middle_channels=64, out_channels=1,
                 num_radial=3, num_spherical=2, num_output_layers=3):
        super().__init__()
        self.out_channels = out_channels
        self.cutoff = cutoff
        self.num_layers = num_layers
        self.num_radial, self.num_spherical = num_radial, num_spherical
        act = swish
        self.act = act
        self.tables = BasisTables(num_spherical, num_radial, 'comenet')
        self.emb = EmbeddingBlock(hidden_channels, act)
        self.interaction_blocks = nn.ModuleList([
            SimpleInteractionBlock(hidden_channels, middle_channels, num_radial, num_spherical, num_output_layers,
                                   hidden_channels, act) for _ in range(num_layers)])
        self.lins = nn.ModuleList([Linear(hidden_channels, hidden_channels) for _ in range(num_output_layers)])
        self.lin_out = Linear(hidden_channels, out_channels)
        self.reset_parameters()

    def reset_parameters(self):
        self.emb.reset_parameters()
        for m in self.interaction_blocks:
            m.reset_parameters()
        for lin in self.lins:
            lin.reset_parameters()
        self.lin_out.reset_parameters()

    def geometry(self, pos, g):
        """dist, theta, phi, tau per edge (comenet.py:297-385)."""
        dev = pos.device
        st = _stream()
        E, N = g.E, g.N
        dist = ops.edge_dist(pos, g, 1)
        add = torch.empty(max(E, 1), dtype=torch.float32, device=dev)

        def nearest_two(seg):
            _, a0 = ops.segment_argmin(dist, None, seg, E)
            call('dig3d_comenet_bump', ptr(a0), N, E, float(self.cutoff), ptr(add), ptr(g.cnt_N), st)
            _, a1 = ops.segment_argmin(dist, add, seg, E)
            return a0, a1

        a0, a1 = nearest_two(g.seg_dst)
        b0, b1 = nearest_two(g.seg_src)
        theta = torch.empty(E, dtype=torch.float32, device=dev)
        phi = torch.empty_like(theta)
        tau = torch.empty_like(theta)
        call('dig3d_comenet_geom', ptr(pos), ptr(g.src), ptr(g.dst), E, ptr(a0), ptr(a1), ptr(b0), ptr(b1),
             ptr(theta), ptr(phi), ptr(tau), ptr(g.cnt_E), st)
        return dist, theta, phi, tau

    def features(self, dist, theta, phi, tau):
        zeros, norms, pref = self.tables.on(dist.device)
        ns, nr = self.num_spherical, self.num_radial
        bes = ops.bessel_basis(dist, self.cutoff, ns, nr, zeros, norms, 0)
        feature1 = ops.sph_basis(bes, None, theta, phi, ns, nr, pref, 1)      # torsion_emb(dist, theta, phi)
        feature2 = ops.sph_basis(bes, None, tau, None, ns, nr, pref, 1)       # angle_emb(dist, tau)
        return feature1, feature2

    needs_triplets = False          # dig_amd/graphed.py: the radius graph without triplet lists

    def _fused_ok(self):
        return True

    def _forward(self, data, g=None):
        z = data.z.long()
        pos = data.pos.contiguous()
        if g is None:
            g = build_graph(pos, data.batch, self.cutoff, triplets=False)
        # (g given: the padded, prebuilt graph of a replayed step.  Everything below is CSR driven or takes the live counts:
        # padded edges get dist = 1 and zero angles (finite features nobody reads), padded nodes have empty segments, their
        # rows are finite and receive zero gradients, GraphNorm writes them as zeros)
        dist, theta, phi, tau = self.geometry(pos, g)
        feature1, feature2 = self.features(dist, theta, phi, tau)
        x = self.emb(z)
        # the composed feature weights W2 W1 of every block (two per block) in ONE launch, their factor gradients in one more
        lfs = [(lf, f) for blk in self.interaction_blocks
               for lf, f, conv in ((blk.lin_feature1, feature1, blk.conv1), (blk.lin_feature2, feature2, blk.conv2))
               if conv.fused_features]
        wcs = [None] * (2 * len(self.interaction_blocks))
        if lfs and len(lfs) == len(wcs) <= 16 and all(lf.composable(f) for lf, f in lfs):
            wcs = ops.compose_weights([(lf.lin2.weight, lf.lin1.weight) for lf, _ in lfs])
        for i, block in enumerate(self.interaction_blocks):
            x = block(x, feature1, feature2, g, (wcs[2 * i], wcs[2 * i + 1]))
        y = _wide([x], [[(lin, self.act) for lin in self.lins]]) if (x.size(0) < ops.comenet_group_rows and len(self.lins)) else None
        if y is not None:
            x = y[0]
        else:
            for lin in self.lins:
                x = lin(x, self.act)
        x = self.lin_out(x)
        return ops.segment_sum(x, g.seg_batch)

    def forward(self, batch_data):
        check_z_bounds(batch_data, self.emb.emb.num_embeddings)
        if getattr(batch_data, 'is_static_graph', False):      # dig_amd/graphed.py: padded, prebuilt graph
            return self._forward(batch_data, batch_data)
        return self._forward(batch_data)
