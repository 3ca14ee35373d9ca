"""SphereNet and DimeNet++ on the HIP engine — one implementation, two public classes.

Drop-in for ``dig.threedgraph.method.SphereNet`` (method/spherenet/spherenet.py:228-320) and
``dig.threedgraph.method.DimeNetPP`` (method/dimenetpp/dimenetpp.py:207-293): same constructor keywords and
defaults, same ``forward(batch_data) -> [B, out_channels]``, same ``state_dict`` keys and shapes
 so checkpoints written by ``run`` (run.py:87-93) load either way.

What runs where:
  graph, dist/angle/torsion, Bessel x harmonics basis ........ HIP (csrc/graph|geometry|basis.hip)
  x_kj[idx_kj] * sbf * t -> scatter (spherenet.py:165-171) ... one fused HIP segment kernel, fwd + bwd
  edge->node / node->graph scatter_add ........................ HIP segment sums (no atomics)
  dense hidden-channel Linears + bias + swish (+ residual) ... f32-MFMA kernels (csrc/dense.hip), fwd + bwd
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...graph import build_graph
from .basis import BasisTables
from ..data import check_z_bounds
from .inits import glorot_orthogonal_


def swish(x):
    return F.silu(x)


def _dense(lin, x, act=None, res=None):
    """act(lin(x)) (+ res): one f32-MFMA kernel (csrc/dense.hip) when act is swish / None, torch otherwise."""
    if act is None or act is swish:
        return ops.linear(x, lin.weight, lin.bias, ops.ACT_SWISH if act is swish else ops.ACT_NONE, res)
    y = act(F.linear(x, lin.weight, lin.bias))
    return y if res is None else res + y


# --------------------------------------------------------------------------------------------- basis
class _DistEmb(nn.Module):
    """Envelope(d/c) * sin(freq * d/c), freq learnable, init pi*[1..nr] (spherenet/features.py:149-182)."""

    def __init__(self, num_radial, cutoff, envelope_exponent):
        super().__init__()
        self.cutoff = cutoff
        self.p = envelope_exponent + 1
        self.freq = nn.Parameter(torch.empty(num_radial))
        self.reset_parameters()

    def reset_parameters(self):
        self.freq.data = torch.arange(1, self.freq.numel() + 1).float().mul_(math.pi)

    def forward(self, dist, cnt=None):
        """one HIP kernel forward, one backward (d and freq gradients), one for the double backward of the force
        path (csrc/diffgeom.hip:k_distemb_*); rows >= cnt (padding of a static-shape batch) are zero."""
        from ... import diffops
        return diffops.dist_emb(dist, self.freq, self.cutoff, self.p, cnt)


def _mul(a, b):
    """a * b; on the energy_and_force route the twice-differentiable three-kernel product (diffops.mul2)"""
    if ops._twice_differentiable:
        from ... import diffops
        return diffops.mul2(a, b)
    return a * b


class _Emb(nn.Module):
    """``emb`` of the reference (spherenet.py:17-32 / dimenetpp.py:20-33): returns (rbf, sbf[, tbf])."""

    def __init__(self, num_spherical, num_radial, cutoff, envelope_exponent, torsion):
        super().__init__()
        self.dist_emb = _DistEmb(num_radial, cutoff, envelope_exponent)
        self.ns, self.nr, self.cutoff, self.torsion = num_spherical, num_radial, cutoff, torsion
        # DimeNet++ multiplies the Bessel part by the envelope (dimenetpp/features.py:214); SphereNet's
        # angle/torsion embeddings have it commented out (spherenet/features.py:193,216).
        self.env_p = 0 if torsion else envelope_exponent + 1
        self.tables = BasisTables(num_spherical, num_radial, 'spherenet')

    def reset_parameters(self):
        self.dist_emb.reset_parameters()

    def edge_front(self, pos, g):
        """(dist, rbf, bes) of the energy route in one launch (diffops.edge_front) instead of three"""
        from ... import diffops
        zeros, norms, _ = self.tables.on(pos.device)
        de = self.dist_emb
        return diffops.edge_front(pos, de.freq, g, 0, de.cutoff, de.p, self.cutoff, self.ns, self.nr, zeros, norms, self.env_p)

    def forward_projected(self, dist, angle, torsion, g, layers, rbf_bes=None):
        """(rbf, Ps, Pt): the basis rows are never materialised; the first basis Linear of every layer is applied
        while they are in registers (csrc/triplet.hip:k_basis_project)."""
        zeros, norms, pref = self.tables.on(dist.device)
        if rbf_bes is not None:
            rbf, bes = rbf_bes
        else:
            rbf = self.dist_emb(dist, g.cnt_E)
            bes = ops.bessel_basis(dist, self.cutoff, self.ns, self.nr, zeros, norms, self.env_p)
        Ps, Pt = ops.basis_project(bes, angle, torsion if self.torsion else None, g.kj, pref, self.ns, self.nr,
                                   [m.lin_sbf1.weight for m in layers],
                                   [m.lin_t1.weight for m in layers] if self.torsion else None, cnt=g.cnt_T)
        return rbf, Ps, Pt

    def forward(self, dist, angle, torsion, g):
        zeros, norms, pref = self.tables.on(dist.device)
        rbf = self.dist_emb(dist, g.cnt_E)
        bes = ops.bessel_basis(dist, self.cutoff, self.ns, self.nr, zeros, norms, self.env_p)
        sbf = ops.sph_basis(bes, g.kj, angle, None, self.ns, self.nr, pref, 0)
        if not self.torsion:
            return rbf, sbf
        tbf = ops.sph_basis(bes, g.kj, angle, torsion, self.ns, self.nr, pref, 0)
        return rbf, sbf, tbf


# --------------------------------------------------------------------------------------------- blocks
class _Residual(nn.Module):
    def __init__(self, hidden, act):
        super().__init__()
        self.act = act
        self.lin1 = nn.Linear(hidden, hidden)
        self.lin2 = nn.Linear(hidden, hidden)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.lin1, self.lin2):
            glorot_orthogonal_(lin.weight, 2.0)
            lin.bias.data.zero_()

    def forward(self, x):
        return _dense(self.lin2, _dense(self.lin1, x, self.act), self.act, res=x)


class _EdgeInit(nn.Module):
    """``init`` (spherenet.py:53-91, dimenetpp.py:55-78)."""

    def __init__(self, num_radial, hidden, act, use_node_features=True, use_extra_node_feature=False):
        super().__init__()
        self.act = act
        self.use_node_features = use_node_features
        self.use_extra_node_feature = use_extra_node_feature
        if use_node_features:
            self.emb = nn.Embedding(95, hidden)
        else:
            self.node_embedding = nn.Parameter(torch.empty(hidden))
            nn.init.normal_(self.node_embedding)
        self.lin_rbf_0 = nn.Linear(num_radial, hidden)
        self.lin = nn.Linear((5 if use_extra_node_feature else 3) * hidden, hidden)
        self.lin_rbf_1 = nn.Linear(num_radial, hidden, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        if self.use_node_features:
            self.emb.weight.data.uniform_(-math.sqrt(3), math.sqrt(3))
        self.lin_rbf_0.reset_parameters()
        self.lin.reset_parameters()
        glorot_orthogonal_(self.lin_rbf_1.weight, 2.0)

    fused_embedding = True

    def forward(self, z, node_feature, rbf, g, factors=False, rb=None, rbf1=None, r1=None):
        # rbf1: a second alias of rbf for lin_rbf_1 (diffops.fan_out: the gradients of all consumers of rbf meet in one launch)
        # rb: (lin_rbf_0 + act, lin_rbf_1) already evaluated by the radial bundle launch (csrc/radial.hip)
        rbf0 = rb[0] if rb is not None else _dense(self.lin_rbf_0, rbf, self.act)
        if (self.use_node_features and self.fused_embedding and not (node_feature is not None and self.use_extra_node_feature)
                and ops.edge_cat_emb_supported(z, self.emb.weight, rbf0, g.seg_dst, g.seg_src)):
            # the embedding lookup inside the edge_cat launch: x = emb(z) is only ever read through x[i], x[j]
            cat = ops.edge_cat_emb(z, self.emb.weight, rbf0, g.seg_dst, g.seg_src)
        else:
            if self.use_node_features:
                x = ops.embedding(z, self.emb.weight)
            else:
                x = self.node_embedding[None, :].expand(z.shape[0], -1)
            if node_feature is not None and self.use_extra_node_feature:
                x = torch.cat((x, node_feature), 1)
            cat = ops.edge_cat(x, rbf0, g.seg_dst, g.seg_src)
        e1 = _dense(self.lin, cat, self.act)                                             # cat([x_i, x_j, rbf0], -1)
        # r1 given: lin_rbf_1(rbf) already evaluated as a head of the force route's radial family (diffops.radial2)
        if r1 is None:
            r1 = rb[1] if rb is not None else _dense(self.lin_rbf_1, rbf if rbf1 is None else rbf1)
        if factors:                                   # (e1, lin_rbf_1(rbf)): e2 is their product (grouped readout)
            return e1, r1
        return e1, _mul(r1, e1)


class _EdgeUpdate(nn.Module):
    """``update_e`` (spherenet.py:94-182, dimenetpp.py:81-161)."""

    def __init__(self, hidden, int_emb, basis_dist, basis_angle, basis_torsion, ns, nr, n_before, n_after,
                 act, torsion):
        super().__init__()
        self.act = act
        self.torsion = torsion
        self.lin_rbf1 = nn.Linear(nr, basis_dist, bias=False)
        self.lin_rbf2 = nn.Linear(basis_dist, hidden, bias=False)
        self.lin_sbf1 = nn.Linear(ns * nr, basis_angle, bias=False)
        self.lin_sbf2 = nn.Linear(basis_angle, int_emb, bias=False)
        if torsion:
            self.lin_t1 = nn.Linear(ns * ns * nr, basis_torsion, bias=False)
            self.lin_t2 = nn.Linear(basis_torsion, int_emb, bias=False)
        self.lin_rbf = nn.Linear(nr, hidden, bias=False)
        self.lin_kj = nn.Linear(hidden, hidden)
        self.lin_ji = nn.Linear(hidden, hidden)
        self.lin_down = nn.Linear(hidden, int_emb, bias=False)
        self.lin_up = nn.Linear(int_emb, hidden, bias=False)
        self.layers_before_skip = nn.ModuleList([_Residual(hidden, act) for _ in range(n_before)])
        self.lin = nn.Linear(hidden, hidden)
        self.layers_after_skip = nn.ModuleList([_Residual(hidden, act) for _ in range(n_after)])
        self.reset_parameters()

    def reset_parameters(self):
        names = ['lin_rbf1', 'lin_rbf2', 'lin_sbf1', 'lin_sbf2', 'lin_kj', 'lin_ji', 'lin_down', 'lin_up',
                 'lin', 'lin_rbf'] + (['lin_t1', 'lin_t2'] if self.torsion else [])
        for n in names:
            m = getattr(self, n)
            glorot_orthogonal_(m.weight, 2.0)
            if m.bias is not None:
                m.bias.data.zero_()
        for r in list(self.layers_before_skip) + list(self.layers_after_skip):
            r.reset_parameters()

    fused_front = True

    def pack_list(self):
        """weights of the front (lin_ji, lin_kj, lin_down) and of the post-aggregation chain, in the order the kernels
        take them: ``_DimeFamily._forward`` packs those of ALL blocks in one launch (ops.pack_weights)."""
        ws = [self.lin_ji.weight, self.lin_kj.weight, self.lin_down.weight, self.lin_up.weight]
        for r in self.layers_before_skip:
            ws += [r.lin1.weight, r.lin2.weight]
        ws.append(self.lin.weight)
        for r in self.layers_after_skip:
            ws += [r.lin1.weight, r.lin2.weight]
        return ws

    def forward(self, e, emb, g, proj=None, factors=False, rb=None, x1_alias=None, packed=None, wc=None, proj2=None):
        """``x1_alias`` (a list, grouped-readout route): receives an alias of x1 that the caller hands to x1's remaining
        consumer (the readout pair of the previous block), so that consumer's gradient reaches ops._Front.backward as
        an argument instead of through a framework addition."""
        rbf0 = emb[0]
        x1, _ = e
        if (self.fused_front and rb is not None and self.act is swish
                and ops.front_supported(x1, rb[0], self.lin_ji, self.lin_kj, self.lin_down)):
            # lin_ji, lin_kj, the product with the radial projection and lin_down in ONE launch per pass
            x_ji, x_kj, x1_skip, x1_ro = ops.front(x1, rb[0], self.lin_ji, self.lin_kj, self.lin_down,
                                                   packed[:, :3] if packed is not None else None)
            if x1_alias is not None:
                x1_alias.append(x1_ro)
            if proj is not None:
                x_kj = ops.triplet_interaction(x_kj, proj[0], proj[1], self.lin_sbf2.weight,
                                               self.lin_t2.weight if self.torsion else None, g)
            else:
                w_sbf = _dense(self.lin_sbf2, _dense(self.lin_sbf1, emb[1]))
                w_t = _dense(self.lin_t2, _dense(self.lin_t1, emb[2])) if self.torsion else None
                x_kj = ops.gather_mul_segment_sum(x_kj, w_sbf, w_t, g.seg_kj, g.seg_ji, composite=g.composite)
            h = self._post_chain(x_kj, x_ji, x1_skip, packed[:, 3:] if packed is not None else None)
            r = rb[1]
            return (h, r) if factors else (h, r * h)
        if ops._twice_differentiable and ops.force_front2 and rb is not None and self.act is swish:
            from ... import diffops
            if diffops.front2_supported(x1, rb[0], self.lin_ji, self.lin_kj, self.lin_down):
                # force route: the whole front as ONE twice-differentiable launch per pass (diffops.front2)
                x_ji, x_kj, x1_skip, x1_ro = diffops.front2(x1, rb[0], self.lin_ji, self.lin_kj, self.lin_down,
                                                            packed[:, :3] if packed is not None else None)
                if x1_alias is not None:          # the caller forms the previous block's e2 from this alias of x1
                    x1_alias.append(x1_ro)
                if (not self.torsion and ops.force_trip2
                        and diffops.trip2_shapes_ok(x_kj, self.lin_sbf1.out_features, self.lin_sbf2.weight)):
                    P = proj2 if proj2 is not None else ops.linear(emb[1], self.lin_sbf1.weight)
                    x_kj = diffops.trip2(x_kj, P, self.lin_sbf2.weight, g)
                else:
                    w_sbf = ops.linear(emb[1], wc[1] if (wc is not None and wc[1] is not None)
                                       else ops.matmul_nn(self.lin_sbf2.weight, self.lin_sbf1.weight))
                    w_t = _dense(self.lin_t2, _dense(self.lin_t1, emb[2])) if self.torsion else None
                    x_kj = ops.gather_mul_segment_sum(x_kj, w_sbf, w_t, g.seg_kj, g.seg_ji, composite=g.composite)
                h = self._post_chain(x_kj, x_ji, x1_skip, packed[:, 3:] if packed is not None else None)
                r = rb[1]
                return (h, r) if factors else (h, _mul(r, h))
        if (ops._twice_differentiable and ops.force_group_front and self.act is swish and x1.is_cuda and x1.dim() == 2
                and x1.size(0) > 0 and self.lin_ji.weight.shape == self.lin_kj.weight.shape and self.lin_ji.out_features > 64
                and self.lin_ji.out_features % 8 == 0 and self.lin_ji.in_features % 4 == 0
                and self.lin_ji.bias is not None and self.lin_kj.bias is not None):
            # force route: lin_ji and lin_kj read the same x1 — ONE grouped twice-differentiable launch per pass
            from ... import diffops
            x_ji, x_kj = diffops.grouped_linear2([x1, x1], [self.lin_ji.weight, self.lin_kj.weight],
                                                 [self.lin_ji.bias, self.lin_kj.bias], ops.ACT_SWISH)
        else:
            x_ji = _dense(self.lin_ji, x1, self.act)
            x_kj = _dense(self.lin_kj, x1, self.act)
        # rb: (lin_rbf2(lin_rbf1(rbf)), lin_rbf(rbf)) already evaluated by the radial bundle launch
        if rb is not None:
            x_kj = _mul(x_kj, rb[0])
        elif ops._twice_differentiable:
            # force route: the two bias-free Linears have no activation between them (spherenet.py:153-155), so
            # they are applied as ONE layer with W2 W1 (a 128x8x6 product) — one set of E-row launches per pass
            # instead of two; the factor gradients follow from the tiny product by autograd
            x_kj = _mul(x_kj, ops.linear(rbf0, wc[0] if wc is not None else ops.matmul_nn(self.lin_rbf2.weight, self.lin_rbf1.weight)))
        else:
            x_kj = x_kj * _dense(self.lin_rbf2, _dense(self.lin_rbf1, rbf0))
        x_kj = _dense(self.lin_down, x_kj, self.act)
        if proj is not None:
            # lin_sbf2 / lin_t2 + gather + products + scatter in ONE kernel; proj = this layer's (Ps, Pt)
            x_kj = ops.triplet_interaction(x_kj, proj[0], proj[1], self.lin_sbf2.weight,
                                           self.lin_t2.weight if self.torsion else None, g)
        else:
            from ... import diffops
            if (ops._twice_differentiable and not self.torsion and ops.force_trip2
                    and diffops.trip2_shapes_ok(x_kj, self.lin_sbf1.out_features, self.lin_sbf2.weight)):
                # force route without torsion (DimeNet++): P = lin_sbf1(sbf) [T, 8] and the fused triplet kernels as a
                # family closed under differentiation (dig_amd/diffops.py:trip2) — no [T, int_emb] tensor in any pass
                # (P handed in: lin_sbf1 of ALL blocks applied as one stacked T-row layer by _DimeFamily._forward)
                P = proj2 if proj2 is not None else ops.linear(emb[1], self.lin_sbf1.weight)
                x_kj = diffops.trip2(x_kj, P, self.lin_sbf2.weight, g)
                h = self._post_chain(x_kj, x_ji, x1)
                r = rb[1] if rb is not None else _dense(self.lin_rbf, rbf0)
                return (h, r) if factors else (h, _mul(r, h))
            if ops._twice_differentiable:       # force route: lin_sbf2 lin_sbf1 as one T-row layer (see above)
                w_sbf = ops.linear(emb[1], wc[1] if (wc is not None and wc[1] is not None)
                                   else ops.matmul_nn(self.lin_sbf2.weight, self.lin_sbf1.weight))
                # (the torsion basis has ns^2 nr = 294 columns: composing would multiply its flops by 6, keep two steps)
                w_t = _dense(self.lin_t2, _dense(self.lin_t1, emb[2])) if self.torsion else None
            else:
                w_sbf = _dense(self.lin_sbf2, _dense(self.lin_sbf1, emb[1]))
                w_t = _dense(self.lin_t2, _dense(self.lin_t1, emb[2])) if self.torsion else None
            # x_kj[idx_kj] * sbf (* t) -> scatter over idx_ji : one fused kernel
            x_kj = ops.gather_mul_segment_sum(x_kj, w_sbf, w_t, g.seg_kj, g.seg_ji, composite=g.composite)
        h = self._post_chain(x_kj, x_ji, x1)
        r = rb[1] if rb is not None else _dense(self.lin_rbf, rbf0)
        if factors:                                   # (h, lin_rbf(rbf0)): e2 = their product, formed by the readout
            return h, r
        return h, _mul(r, h)

    fused_chain = True

    def _post_chain(self, x_kj, x_ji, x1, packed=None):
        """lin_up + skip, residual layers, lin + skip, residual layers (spherenet.py:172-182) — ONE forward launch
        when the shapes fit the chain kernel (hidden = 128), else layer by layer."""
        if self.fused_chain and self.act is swish:
            A = ops.ACT_SWISH
            layers = [(self.lin_up.weight, None, A, 1, x_ji, True)]
            for r in self.layers_before_skip:
                layers += [(r.lin1.weight, r.lin1.bias, A, 0, None, False), (r.lin2.weight, r.lin2.bias, A, 2, None, True)]
            layers.append((self.lin.weight, self.lin.bias, A, 1, x1, True))
            for r in self.layers_after_skip:
                layers += [(r.lin1.weight, r.lin1.bias, A, 0, None, False), (r.lin2.weight, r.lin2.bias, A, 2, None, True)]
            if ops.chain_supported(x_kj, layers):
                return ops.chain(x_kj, layers, packed)
            from ... import diffops
            if diffops.chain2_supported(x_kj, layers):          # energy_and_force: the twice-differentiable chain
                return diffops.chain2(x_kj, layers, packed)
        h = _dense(self.lin_up, x_kj, self.act, res=x_ji)
        for layer in self.layers_before_skip:
            h = layer(h)
        h = _dense(self.lin, h, self.act, res=x1)
        for layer in self.layers_after_skip:
            h = layer(h)
        return h


class _NodeOutput(nn.Module):
    """``update_v`` (spherenet.py:185-216, dimenetpp.py:164-195)."""

    def __init__(self, hidden, out_emb, out_channels, n_layers, act, output_init):
        super().__init__()
        self.act = act
        self.output_init = output_init
        self.lin_up = nn.Linear(hidden, out_emb, bias=True)
        self.lins = nn.ModuleList([nn.Linear(out_emb, out_emb) for _ in range(n_layers)])
        self.lin = nn.Linear(out_emb, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_orthogonal_(self.lin_up.weight, 2.0)
        for lin in self.lins:
            glorot_orthogonal_(lin.weight, 2.0)
            lin.bias.data.zero_()
        if self.output_init == 'zeros':
            self.lin.weight.data.zero_()
        if self.output_init == 'GlorotOrthogonal':
            glorot_orthogonal_(self.lin.weight, 2.0)

    def forward(self, e, g):
        v = ops.segment_sum(e[1], g.seg_dst)
        v = _dense(self.lin_up, v)
        for lin in self.lins:
            v = _dense(lin, v, self.act)
        return self.lin(v)


class _GraphSum(nn.Module):
    """``update_u`` (spherenet.py:219-225): u += scatter(v, batch)."""

    def forward(self, u, v, g):
        return u + ops.segment_sum(v, g.seg_batch)


# --------------------------------------------------------------------------------------------- models
class _DimeFamily(nn.Module):
    _torsion = False

    def _build(self, energy_and_force, cutoff, num_layers, hidden_channels, out_channels, int_emb_size,
               basis_dist, basis_angle, basis_torsion, out_emb_channels, num_spherical, num_radial,
               envelope_exponent, num_before_skip, num_after_skip, num_output_layers, act, output_init,
               use_node_features=True, use_extra_node_feature=False, extra_node_feature_dim=1):
        self.cutoff = cutoff
        self.energy_and_force = energy_and_force
        self.use_extra_node_feature = use_extra_node_feature
        if use_extra_node_feature:
            self.extra_emb = nn.Linear(extra_node_feature_dim, hidden_channels)
        self.init_e = _EdgeInit(num_radial, hidden_channels, act, use_node_features, use_extra_node_feature)
        self.init_v = _NodeOutput(hidden_channels, out_emb_channels, out_channels, num_output_layers, act,
                                  output_init)
        self.init_u = _GraphSum()
        self.emb = _Emb(num_spherical, num_radial, cutoff, envelope_exponent, self._torsion)
        self.update_vs = nn.ModuleList([
            _NodeOutput(hidden_channels, out_emb_channels, out_channels, num_output_layers, act, output_init)
            for _ in range(num_layers)])
        self.update_es = nn.ModuleList([
            _EdgeUpdate(hidden_channels, int_emb_size, basis_dist, basis_angle, basis_torsion, num_spherical,
                        num_radial, num_before_skip, num_after_skip, act, self._torsion)
            for _ in range(num_layers)])
        self.update_us = nn.ModuleList([_GraphSum() for _ in range(num_layers)])
        self.reset_parameters()

    fused_triplets = True      # False: basis tables + GEMMs (the unfused route; also the force path)

    def _fused_ok(self):
        if len(self.update_es) == 0:
            return False
        m = self.update_es[0]
        sizes = [m.lin_sbf1.out_features] + ([m.lin_t1.out_features] if self._torsion else [])
        return ops.triplet_fused_supported(m.lin_down.out_features, self.emb.ns, self.emb.nr, sizes, self._torsion)

    def reset_parameters(self):
        if self.use_extra_node_feature:
            self.extra_emb.reset_parameters()
        self.init_e.reset_parameters()
        self.init_v.reset_parameters()
        self.emb.reset_parameters()
        for m in self.update_es:
            m.reset_parameters()
        for m in self.update_vs:
            m.reset_parameters()

    def forward(self, batch_data):
        if self.init_e.use_node_features:
            check_z_bounds(batch_data, self.init_e.emb.num_embeddings)
        extra = None
        if self.use_extra_node_feature and getattr(batch_data, 'node_feature', None) is not None:
            extra = self.extra_emb(batch_data.node_feature)           # spherenet.py:261-262
        if getattr(batch_data, 'is_static_graph', False):      # dig_amd/graphed.py: padded, prebuilt graph
            if self.energy_and_force:                           # pos_leaf: the differentiable alias of the positions
                with ops.composite_mode(True):
                    return self._forward(batch_data.z, batch_data.pos_leaf, None, extra, batch_data)
            return self.forward_graph(batch_data.z, batch_data.pos, batch_data, extra)
        z, pos, batch = batch_data.z, batch_data.pos, batch_data.batch
        if self.energy_and_force:
            pos.requires_grad_()
        with ops.composite_mode(pos.requires_grad):    # forces need a twice-differentiable graph
            return self._forward(z, pos, batch, extra)

    def forward_graph(self, z, pos, g, extra=None):
        """Forward on a prebuilt ``MolGraph`` (possibly padded to a static capacity: dig_amd/graphed.py)."""
        with ops.composite_mode(False):
            return self._forward(z, pos, None, extra, g)

    def _forward(self, z, pos, batch, extra, g=None):
        if g is None:
            g = build_graph(pos, batch, self.cutoff, triplets=True)
        g.composite = bool(pos.requires_grad)
        proj = None
        sbf_ops = None
        if pos.requires_grad:
            from .force_path import dime_geometry_differentiable
            from ... import diffops
            # DimeNet++ on the closed triplet family with stacked first basis Linears: the angular table is never formed
            Lb = len(self.update_es)
            fused_sbf = bool((not self._torsion) and ops.force_trip2 and ops.force_trip2_stacked and ops.force_sbf_fused
                             and Lb > 1 and pos.is_cuda and g.T > 0 and self.grouped_readout
                             and self._readout_ok(pos, [self.init_v] + list(self.update_vs), g, forces=True)
                             and diffops.sbf_project_supported(self.emb.ns, self.emb.nr, Lb,
                                                               [m.lin_sbf1.out_features for m in self.update_es])
                             and all(m.lin_sbf1.weight.is_leaf for m in self.update_es)
                             and self.update_es[0].lin_down.out_features in (16, 32, 64, 128, 256)
                             and all(tuple(m.lin_sbf2.weight.shape) == (m.lin_down.out_features, 8) for m in self.update_es))
            emb = dime_geometry_differentiable(self, pos, g, fused_sbf)
            if fused_sbf:
                sbf_ops, emb = (emb[2], emb[3]), (emb[0], None)
        else:
            posc = pos.contiguous()
            fused = self.fused_triplets and self._fused_ok()
            rbf_bes = None
            if fused and ops.edge_front_fused and posc.is_cuda and posc.dtype == torch.float32 and g.E > 0:
                dist, rbf0, bes = self.emb.edge_front(posc, g)
                rbf_bes = (rbf0, bes)
            else:
                dist = ops.edge_dist(posc, g, 0)
            angle, torsion, _ = ops.triplet_geom(posc, g, self._torsion)
            if fused:
                # (r05, measured and not kept: the basis projection — a 57-us kernel nothing before the first triplet
                # interaction depends on, its weight gradient a leaf of the backward pass — on a SECOND stream, i.e. a
                # parallel branch of the captured graph: 1.556 vs 1.525 ms per config-2 step, 5.431 vs 5.424 config 4, same
                # box; the replayed graph does not overlap its branches, the fork / join only adds dependencies.  r03 saw
                # the same with the next batch's graph build beside the replay.)
                rbf, Ps, Pt = self.emb.forward_projected(dist, angle, torsion, g, self.update_es, rbf_bes)
                emb = (rbf,)
                proj = [(Ps[l], Pt[l] if Pt is not None else None) for l in range(len(self.update_es))]
            else:
                emb = self.emb(dist, angle, torsion, g)
        blocks = [self.init_v] + list(self.update_vs)
        if self.grouped_readout and self._readout_ok(emb[0], blocks, g):
            # the L + 1 output blocks depend only on their layer's e2 = lin_rbf(rbf) * e1: run the edge chain first
            # (keeping the two FACTORS of every e2), then every stage of ALL output blocks as one grouped launch
            # (csrc/readout.hip) — same arithmetic, same summation order, e2 never written
            rb = self._radial_bundle(emb[0])
            e = self.init_e(z, extra, emb[0], g, factors=True, rb=rb[0] if rb else None)
            pairs = [(e[1], e[0])]
            # the weights of every front and chain of this forward in MFMA operand order: one launch for all blocks
            packs, per = None, 0
            if rb is not None and len(self.update_es) and _EdgeUpdate.fused_front and _EdgeUpdate.fused_chain:
                lists = [m.pack_list() for m in self.update_es]
                per = len(lists[0])
                flat = [w for ws in lists for w in ws]
                if ops.packable(flat) and lists[0][0].shape == (128, 128) and not ops._twice_differentiable:
                    packs = ops.pack_weights(flat)
            for l, upd_e in enumerate(self.update_es):
                box = []
                e = upd_e(e, emb, g, proj[l] if proj is not None else None, factors=True, rb=rb[l + 1] if rb else None,
                          x1_alias=box, packed=packs[:, l * per:(l + 1) * per] if packs is not None else None)
                if box:           # the fused front handed back an alias of its x1 for the previous block's readout pair
                    pairs[-1] = (pairs[-1][0], box[0])
                pairs.append((e[1], e[0]))
            return ops.grouped_readout(pairs, blocks, g)
        if (self.grouped_readout and ops._twice_differentiable and self._readout_ok(emb[0], blocks, g, forces=True)):
            # energy_and_force: the same regrouping on the twice-differentiable operator set (dig_amd/diffops.py)
            # rbf has 2 + 2 L consumers (init: lin_rbf_0, lin_rbf_1; per block lin_rbf2 lin_rbf1 and lin_rbf): each gets its own
            # alias, the 2 + 2 L gradients are summed by one launch per backward pass instead of 1 + 2 L framework additions
            from ... import diffops
            Lr = len(self.update_es)
            Hr = self.update_es[0].lin_rbf.out_features if Lr else 0
            # the 2 L radial projections of the blocks as ONE closed family on the matrix-core radial kernels (diffops.radial2):
            # a single consumer of rbf instead of 2 L
            rad2 = bool(ops.force_radial2 and Lr and 0 < 2 * Lr <= 16
                        and emb[0].is_cuda and emb[0].size(0) > 0 and emb[0].dim() == 2 and emb[0].size(1) <= 8
                        and 8 <= Hr <= 256 and Hr % 4 == 0
                        and all(m.lin_rbf.out_features == Hr and m.lin_rbf2.out_features == Hr and m.lin_rbf.bias is None
                                for m in self.update_es))
            # the composed weights lin_rbf2·lin_rbf1 and lin_sbf2·lin_sbf1 of every block (spherenet.py:153-157: two bias-free
            # Linears with nothing between them) in ONE launch, their factor gradients in one more (were 6 library GEMM
            # launches per block and step)
            L = len(self.update_es)
            wcs = None
            # (without torsion the angular pair is not composed: P = lin_sbf1(sbf) feeds the closed fused-triplet family)
            trip2 = (not self._torsion) and ops.force_trip2
            if 0 < 2 * L <= 16 and emb[0].is_cuda and trip2:
                flat = ops.compose_weights([(m.lin_rbf2.weight, m.lin_rbf1.weight) for m in self.update_es])
                wcs = [(flat[l], None) for l in range(L)]
            elif 0 < 2 * L <= 16 and emb[0].is_cuda:
                flat = ops.compose_weights([p for m in self.update_es
                                            for p in ((m.lin_rbf2.weight, m.lin_rbf1.weight), (m.lin_sbf2.weight, m.lin_sbf1.weight))])
                wcs = [(flat[2 * l], flat[2 * l + 1]) for l in range(L)]
            rbs, r1_init = None, None
            if rad2 and wcs is not None:
                Wr = [wcs[l][0] for l in range(L)] + [m.lin_rbf.weight for m in self.update_es]
                # init's lin_rbf_1 (bias-free, no activation: dimenetpp.py:76) is one more head of the same family
                i1 = self.init_e.lin_rbf_1
                with_init = i1.bias is None and i1.out_features == Hr and len(Wr) < 16
                if with_init:
                    Wr = Wr + [i1.weight]
                if diffops.radial2_supported(emb[0], Wr):
                    R = diffops.radial2(emb[0], Wr)
                    rbs = [(R[l], R[L + l]) for l in range(L)]
                    r1_init = R[2 * L] if with_init else None
            rad2 = rbs is not None
            # consumers of rbf: lin_rbf_0 (+ lin_rbf_1) of init and either the one radial family or the 2 L projections
            nfan = (1 if r1_init is not None else 2) + (1 if rad2 else 2 * L)
            rfan = None
            if (not rad2) and ops.force_fan_out and emb[0].is_cuda and emb[0].requires_grad and 3 <= nfan <= 16:
                rfan = diffops.fan_out(emb[0], nfan)
            e = self.init_e(z, extra, rfan[0] if rfan else emb[0], g, factors=True,
                            rbf1=rfan[1] if rfan else None, r1=r1_init)   # (e1, lin_rbf_1(rbf)): its e2 is formed below
            e2s = []
            # trip2 route: P_l = lin_sbf1_l(sbf) of every block as ONE [T, ns*nr] -> [T, 8 L] layer with stacked weights (the
            # table is read once per pass instead of L times), split into the blocks' contiguous [T, 8] operands
            P2 = None
            bs = [m.lin_sbf1.out_features for m in self.update_es]
            if sbf_ops is not None:
                from ... import diffops
                g.zero_trip_tail = False        # the basis kernels never read a row behind the live triplet count
                P2 = diffops.sbf_project(sbf_ops[0], sbf_ops[1], torch.cat([m.lin_sbf1.weight for m in self.update_es], 0), g,
                                         self.emb.ns, self.emb.nr, self.emb.tables.on(pos.device)[2])
            elif trip2 and 1 < L <= 8 and all(b == 8 for b in bs) and emb[1].is_cuda and emb[1].size(0) > 0 and ops.force_trip2_stacked:
                from ... import diffops
                P2 = diffops.split_cols8(ops.linear(emb[1], torch.cat([m.lin_sbf1.weight for m in self.update_es], 0)), L)
            # the 2 L radial projections of the blocks — lin_rbf2 lin_rbf1 (composed) and lin_rbf, all [hidden, num_radial] on
            # the SAME rbf rows — as one grouped twice-differentiable launch per pass instead of 2 L (each E-row launch of a
            # K = 6 layer is ~10-25 us of floor in every one of the four passes)
            H = self.update_es[0].lin_rbf.out_features if L else 0
            if (rbs is None and ops.force_group_radial and wcs is not None and 0 < 2 * L <= 8 and emb[0].is_cuda and emb[0].size(0) > 0
                    and H > 64 and H % 8 == 0 and all(m.lin_rbf.out_features == H and m.lin_rbf2.out_features == H
                                                       and m.lin_rbf.bias is None for m in self.update_es)):
                from ... import diffops
                Wr = [wcs[l][0] for l in range(L)] + [m.lin_rbf.weight for m in self.update_es]
                R = diffops.grouped_linear2(rfan[2:2 + 2 * L] if rfan else [emb[0]] * (2 * L), Wr, [None] * (2 * L), ops.ACT_NONE)
                rbs = [(R[l], R[L + l]) for l in range(L)]
            # every block returns the FACTORS (h, r) of its e2 = r * h; the product is formed once the NEXT block's front has
            # handed back an alias of h (its x1) for it — h's three consumers then meet inside k_front_bwd, not in two
            # framework additions per block and pass
            # the weights of every front and chain in MFMA operand order: ONE launch for all blocks (was one per front and per
            # chain: 2 L launches of ~4.6 us per step)
            packs, per = None, 0
            if (rbs is not None and ops.force_front2 and _EdgeUpdate.fused_chain and all(m.act is swish for m in self.update_es)):
                lists = [m.pack_list() for m in self.update_es]
                per = len(lists[0])
                flat = [w for ws in lists for w in ws]
                if (ops.packable(flat) and all(len(ws) == per for ws in lists)
                        and all(w.shape == (128, 128) for ws in lists for w in (ws[0], ws[1]))
                        and all(w.is_leaf for w in flat)):
                    packs = ops.pack_weights(flat)
            pend = e
            for l, upd_e in enumerate(self.update_es):
                box = []
                e = upd_e(e, emb, g, None, wc=wcs[l] if wcs else None, proj2=P2[l] if P2 is not None else None,
                          rb=rbs[l] if rbs is not None else None, factors=True, x1_alias=box,
                          packed=packs[:, l * per:(l + 1) * per] if packs is not None else None)
                if pend is not None:
                    e2s.append((pend[1], box[0] if box else pend[0]))
                pend = e
                e = (e[0], None)
            if pend is not None:
                e2s.append((pend[1], pend[0]))
            # e2 = r * h of every block: formed inside the grouped edge -> node sum where the shapes allow (diffops.mul_segsum_grouped)
            rs_, hs_ = [p[0] for p in e2s], [p[1] for p in e2s]
            if ops.force_mul_segsum and ops.force_group_segsum and diffops.mul_segsum_grouped_supported(rs_, hs_, g.seg_dst):
                return self._readout_forces(None, blocks, g, vs=diffops.mul_segsum_grouped(rs_, hs_, g.seg_dst))
            return self._readout_forces([_mul(r_, h_) for r_, h_ in e2s], blocks, g)
        e = self.init_e(z, extra, emb[0], g)
        v = self.init_v(e, g)
        u = self.init_u(torch.zeros(g.B, v.size(1), dtype=v.dtype, device=v.device), v, g)
        for l, (upd_e, upd_v, upd_u) in enumerate(zip(self.update_es, self.update_vs, self.update_us)):
            e = upd_e(e, emb, g, proj[l] if proj is not None else None)
            v = upd_v(e, g)
            u = upd_u(u, v, g)
        return u

    grouped_readout = True
    radial_bundle = True

    def _radial_bundle(self, rbf):
        """all 2 + 2L radial projections of the forward (lin_rbf_0 + act, lin_rbf_1; per layer lin_rbf2(lin_rbf1(.)) and
        lin_rbf) in one launch -> [(rbf0_act, r1), (proj_kj_l, r_l) ...], or None when the shapes do not fit."""
        ie = self.init_e
        if not self.radial_bundle or ie.act is not swish or not rbf.is_cuda or rbf.size(0) == 0:
            return None
        H = ie.lin_rbf_0.out_features
        heads = [(H, None), (H, None)]
        for m in self.update_es:
            heads += [(m.lin_rbf2.out_features, m.lin_rbf1.out_features), (m.lin_rbf.out_features, None)]
        if not ops.radial_bundle_supported(rbf.size(1), heads):
            return None
        spec = [('single', ie.lin_rbf_0.weight, ie.lin_rbf_0.bias, ops.ACT_SWISH), ('single', ie.lin_rbf_1.weight, None, ops.ACT_NONE)]
        for m in self.update_es:
            spec += [('two', m.lin_rbf1.weight, m.lin_rbf2.weight), ('single', m.lin_rbf.weight, None, ops.ACT_NONE)]
        out = ops.radial_bundle(rbf, spec)
        return [(out[2 * k], out[2 * k + 1]) for k in range(len(self.update_es) + 1)]

    def _readout_ok(self, rbf, blocks, g, forces=False):
        b0 = blocks[0]
        ok = (b0.act is swish and rbf.is_cuda and rbf.dtype == torch.float32 and g.E > 0 and g.N > 0
              and getattr(g, '_sorted_edges', True))
        if forces:      # grouped MFMA kernels of the second-order route need N > 64 outputs per layer
            return (ok and 1 <= len(blocks) <= 8 and b0.lin_up.out_features % 8 == 0 and b0.lin_up.out_features > 64
                    and b0.lin_up.in_features % 4 == 0)
        return ok and ops.grouped_readout_supported(b0.lin_up.in_features, b0.lin_up.out_features, b0.lin.out_features,
                                                    len(blocks))

    def _readout_forces(self, e2s, blocks, g, vs=None):
        """output blocks of all layers on the twice-differentiable kernels: segment sums and heads per block (linear
        maps, closed under differentiation), the four dense stages of ALL blocks as one grouped launch each.  ``vs``: the
        edge -> node sums, already formed."""
        from ... import diffops
        if vs is not None:
            pass
        elif ops.force_group_segsum and diffops.segsum_grouped_supported(e2s, g.seg_dst):
            vs = diffops.segsum_grouped(e2s, g.seg_dst)          # one launch per pass for all L + 1 blocks
        else:
            vs = [ops.segment_sum(e2, g.seg_dst) for e2 in e2s]
        wl = [[(b.lin_up.weight, b.lin_up.bias, ops.ACT_NONE, 0)] + [(lin.weight, lin.bias, ops.ACT_SWISH, 0) for lin in b.lins]
              for b in blocks]
        if ops.force_wide2 and diffops.wide2_supported(vs, wl):
            hs = diffops.wide2(vs, wl)          # lin_up + lins of ALL blocks: one launch per pass (csrc/wide.hip)
        else:
            hs = diffops.grouped_linear2(vs, [b.lin_up.weight for b in blocks], [b.lin_up.bias for b in blocks], ops.ACT_NONE)
            for j in range(len(blocks[0].lins)):
                hs = diffops.grouped_linear2(hs, [b.lins[j].weight for b in blocks], [b.lins[j].bias for b in blocks],
                                             ops.ACT_SWISH)
        Wl = [b.lin.weight for b in blocks]
        if self.grouped_heads and all(b.lin.bias is None for b in blocks) and diffops.heads2_supported(hs, Wl):
            # all heads as row dot products in one launch per pass, all graph sums in one (reference accumulation order)
            return ops.graph_sum_group(diffops.heads2(hs, Wl), g.seg_batch)
        u = None
        for b, h in zip(blocks, hs):
            y = ops.segment_sum(b.lin(h), g.seg_batch)
            u = y if u is None else u + y
        return u

    grouped_heads = True      # False: per-block torch heads (tests compare the routes)


class SphereNet(_DimeFamily):
    r"""SphereNet (`"Spherical Message Passing for 3D Molecular Graphs"`), API of
    dig/threedgraph/method/spherenet/spherenet.py:253-259 (same keywords and defaults)."""
    _torsion = True

    def __init__(self, energy_and_force=False, cutoff=5.0, num_layers=4, hidden_channels=128, out_channels=1,
                 int_emb_size=64, basis_emb_size_dist=8, basis_emb_size_angle=8, basis_emb_size_torsion=8,
                 out_emb_channels=256, num_spherical=7, num_radial=6, envelope_exponent=5, num_before_skip=1,
                 num_after_skip=2, num_output_layers=3, act=swish, output_init='GlorotOrthogonal',
                 use_node_features=True, use_extra_node_feature=False, extra_node_feature_dim=1):
        super().__init__()
        self._build(energy_and_force, cutoff, num_layers, hidden_channels, out_channels, int_emb_size,
                    basis_emb_size_dist, basis_emb_size_angle, basis_emb_size_torsion, out_emb_channels,
                    num_spherical, num_radial, envelope_exponent, num_before_skip, num_after_skip,
                    num_output_layers, act, output_init, use_node_features, use_extra_node_feature,
                    extra_node_feature_dim)


class DimeNetPP(_DimeFamily):
    r"""DimeNet++ under the 3DGN framework, API of dig/threedgraph/method/dimenetpp/dimenetpp.py:230-235."""
    _torsion = False

    def __init__(self, energy_and_force=False, cutoff=5.0, num_layers=4, hidden_channels=128, out_channels=1,
                 int_emb_size=64, basis_emb_size=8, out_emb_channels=256, num_spherical=7, num_radial=6,
                 envelope_exponent=5, num_before_skip=1, num_after_skip=2, num_output_layers=3, act=swish,
                 output_init='GlorotOrthogonal'):
        super().__init__()
        self._build(energy_and_force, cutoff, num_layers, hidden_channels, out_channels, int_emb_size,
                    basis_emb_size, basis_emb_size, basis_emb_size, out_emb_channels, num_spherical, num_radial,
                    envelope_exponent, num_before_skip, num_after_skip, num_output_layers, act, output_init)
