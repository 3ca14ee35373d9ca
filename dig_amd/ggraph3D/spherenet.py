"""Drop-in for ``dig.ggraph3D.method.G_SphereNet.model.spherenet.SphereNet`` (spherenet.py:221-297 there): same
constructor signature, ``forward(z, pos, batch) -> [N, hidden]`` node embeddings, ``dist_only_forward``, same
``state_dict`` layout.  Differences from threedgraph's SphereNet, all reproduced here:

  * torsion is taken against the NEAREST node of j (knn, no cutoff; second nearest if that is i) —
    geometric_computing.py:13-19,83-103 -> csrc/geometry.hip:k_nearest_two / k_triplet_geom_knn;
  * ``update_e`` / ``update_v`` / the final node update replace "isolated" rows by their input through
    ``x + scatter(y[idx] - x[idx], idx, reduce='mean')`` (:171-172,205,297) -> the fused segment-mean kernels;
  * the output blocks end in ``Linear(out_emb, hidden)`` with bias and return node features, there is no graph readout.
"""
from math import sqrt

import torch
from torch import nn

from .. import ops
from .._hip import call, ptr
from ..graph import build_graph, _stream
from ..threedgraph.method.dime_family import _Emb, _EdgeUpdate, _dense, swish
from ..threedgraph.method.inits import glorot_orthogonal_


def _mean_fix(y, x, seg):
    """x + scatter(y[idx] - x[idx], idx, dim_size=len(x), reduce='mean') for the index behind ``seg``."""
    d = ops.gather_rows(y, seg) - ops.gather_rows(x, seg)
    return x + ops.segment_mean(d, seg)


class _Init(nn.Module):
    """``init`` (spherenet.py:60-82): returns ((e1, e2), node type embedding)."""

    def __init__(self, num_node_types, num_radial, hidden, act):
        super().__init__()
        self.act = act
        self.emb = nn.Embedding(num_node_types, hidden)
        self.lin_rbf_0 = nn.Linear(num_radial, hidden)
        self.lin = nn.Linear(3 * hidden, hidden)
        self.lin_rbf_1 = nn.Linear(num_radial, hidden, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        self.emb.weight.data.uniform_(-sqrt(3), sqrt(3))
        self.lin_rbf_0.reset_parameters()
        self.lin.reset_parameters()
        glorot_orthogonal_(self.lin_rbf_1.weight, 2.0)

    def forward(self, z, rbf, g):
        x = self.emb(z)
        rbf0 = _dense(self.lin_rbf_0, rbf, self.act)
        e1 = _dense(self.lin, torch.cat([ops.gather_rows(x, g.seg_dst), ops.gather_rows(x, g.seg_src), rbf0], dim=-1),
                    self.act)
        return (e1, _dense(self.lin_rbf_1, rbf) * e1), x


class _UpdateE(_EdgeUpdate):
    """``update_e`` (spherenet.py:85-174) = threedgraph's block + the mean fix over the edges that occur in a triplet."""

    def forward(self, e, emb, g, proj=None):
        x1, x2 = e
        e1, e2 = super().forward(e, emb, g, proj)
        seg = g.seg_non_iso
        return _mean_fix(e1, x1, seg), _mean_fix(e2, x2, seg)


class _UpdateV(nn.Module):
    """``update_v`` (spherenet.py:177-206)."""

    def __init__(self, hidden, out_emb, num_output_layers, act):
        super().__init__()
        self.act = act
        self.lin_up = nn.Linear(hidden, out_emb, bias=True)
        self.lins = nn.ModuleList([nn.Linear(out_emb, out_emb) for _ in range(num_output_layers - 1)])
        self.lin = nn.Linear(out_emb, hidden)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_orthogonal_(self.lin_up.weight, 2.0)
        for lin in self.lins:
            glorot_orthogonal_(lin.weight, 2.0)
            lin.bias.data.fill_(0)
        glorot_orthogonal_(self.lin.weight, 2.0)
        self.lin.bias.data.fill_(0)

    def forward(self, e, g):
        v = ops.segment_sum(e[1], g.seg_dst)
        v = _dense(self.lin_up, v)
        for lin in self.lins:
            v = _dense(lin, v, self.act)
        v = _dense(self.lin, v)
        return ops.segment_mean(ops.gather_rows(v, g.seg_dst), g.seg_dst)       # scatter(v[i], i, reduce='mean')


class SphereNet(nn.Module):
    def __init__(self, cutoff, num_node_types, num_layers, hidden_channels, int_emb_size, basis_emb_size,
                 out_emb_channels, num_spherical, num_radial, envelope_exponent=5, num_before_skip=1, num_after_skip=2,
                 num_output_layers=3, act=swish):
        super().__init__()
        self.cutoff = cutoff
        self.init_e = _Init(num_node_types, num_radial, hidden_channels, act)
        self.init_v = _UpdateV(hidden_channels, out_emb_channels, num_output_layers, act)
        self.emb = _Emb(num_spherical, num_radial, cutoff, envelope_exponent, True)
        self.update_vs = nn.ModuleList([_UpdateV(hidden_channels, out_emb_channels, num_output_layers, act)
                                        for _ in range(num_layers)])
        self.update_es = nn.ModuleList([
            _UpdateE(hidden_channels, int_emb_size, basis_emb_size, basis_emb_size, basis_emb_size, num_spherical,
                     num_radial, num_before_skip, num_after_skip, act, True) for _ in range(num_layers)])
        self.reset_parameters()

    def reset_parameters(self):
        self.init_e.reset_parameters()
        self.init_v.reset_parameters()
        self.emb.reset_parameters()
        for m in self.update_es:
            m.reset_parameters()
        for m in self.update_vs:
            m.reset_parameters()

    def dist_only_forward(self, z, pos, batch):
        g = build_graph(pos, batch, self.cutoff, triplets=False)
        dist = ops.edge_dist(pos.contiguous(), g, 0)
        e, _ = self.init_e(z, self.emb.dist_emb(dist), g)
        v = self.init_v(e, g)
        for l in range(len(self.update_es)):
            v = self.update_vs[l](e, g)
        return v

    def geometry(self, pos, batch, g):
        """dist, angle, torsion with the knn torsion reference (geometric_computing.py:57-103)."""
        dev = pos.device
        n1 = torch.empty(g.N, dtype=torch.int32, device=dev)
        n2 = torch.empty(g.N, dtype=torch.int32, device=dev)
        st = _stream()
        call('dig3d_nearest_two', ptr(pos), ptr(g.batch32), ptr(g.ptr), g.N, ptr(n1), ptr(n2), st)
        angle = torch.empty(g.T, dtype=torch.float32, device=dev)
        torsion = torch.empty(g.T, dtype=torch.float32, device=dev)
        call('dig3d_triplet_geom_knn', ptr(pos), ptr(g.src), ptr(g.dst), ptr(g.kj), ptr(g.ji), g.T, ptr(n1), ptr(n2),
             ptr(angle), ptr(torsion), st)
        return ops.edge_dist(pos, g, 0), angle, torsion

    def forward(self, z, pos, batch):
        pos = pos.contiguous()
        g = build_graph(pos, batch, self.cutoff, triplets=True)
        g.composite = False
        # edges that occur in a triplet, as ji or as kj (spherenet.py:170: non_iso_idx = cat(idx_ji, idx_kj))
        g.seg_non_iso = ops._seg_from_index(torch.cat(g.idx_kj_ji[::-1]), g.E)
        dist, angle, torsion = self.geometry(pos, batch, g)
        emb = self.emb(dist, angle, torsion, g)                      # (rbf, sbf, tbf) tables
        e, node_type_emb = self.init_e(z, emb[0], g)
        v = self.init_v(e, g)
        for upd_e, upd_v in zip(self.update_es, self.update_vs):
            e = upd_e(e, emb, g)
            v = upd_v(e, g)
        return _mean_fix(v, node_type_emb, g.seg_src)                # scatter(v[j] - emb[j], j, reduce='mean')
