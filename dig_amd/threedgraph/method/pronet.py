"""ProNet on the HIP engine.  Drop-in for ``dig.threedgraph.method.ProNet`` (method/pronet/pronet.py:256-473): same
constructor keywords/defaults, ``forward(batch_data)`` reading ``x, coords_ca, coords_n, coords_c, bb_embs,
side_chain_embs, batch``, and the same ``state_dict`` layout.

HIP: radius graph with the model's ``max_num_neighbors`` (:386), the per-edge geometry relative to the sequence
neighbours and the Euler angles between residue frames (:392-446, csrc/geometry.hip:k_pronet_geom), the positional
embedding (:362-372), the Bessel x harmonics features (pronet/features.py = comenet/features.py), every
``EdgeGraphConv`` message + aggregation (``torch_sparse.matmul`` / ``propagate`` at :111-147) as the fused
gather-multiply-segment-sum kernel, dense Linears + swish on the f32-MFMA kernels.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ..._hip import call, ptr
from ...graph import build_graph, _stream
from .basis import BasisTables
from .inits import glorot_

num_aa_type = 26
num_side_chain_embs = 8
num_bb_embs = 6


def swish(x):
    return x * torch.sigmoid(x)


class Linear(nn.Module):
    """pronet.py:29-69 — glorot (or zeros) weight, zero bias."""

    def __init__(self, in_channels, out_channels, bias=True, weight_initializer='glorot'):
        super().__init__()
        self.in_channels, self.out_channels, self.weight_initializer = in_channels, out_channels, weight_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == 'glorot':
            glorot_(self.weight)
        elif self.weight_initializer == 'zeros':
            self.weight.data.zero_()
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, act=None, res=None):
        return ops.linear(x, self.weight, self.bias, ops.ACT_SWISH if act is swish else ops.ACT_NONE, res)


class TwoLinear(nn.Module):
    def __init__(self, in_channels, middle_channels, out_channels, bias=False, act=False):
        super().__init__()
        self.lin1 = Linear(in_channels, middle_channels, bias=bias)
        self.lin2 = Linear(middle_channels, out_channels, bias=bias)
        self.act = act

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def forward(self, x):
        a = swish if self.act else None
        return self.lin2(self.lin1(x, a), a)


class EdgeGraphConv(nn.Module):
    """pronet.py:111-147: lin_l(sum_{j->i} edge_weight * x_j) + lin_r(x_i)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = Linear(in_channels, out_channels)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, g, edge_weight):
        agg = ops.gather_mul_segment_sum(x, edge_weight, None, g.seg_src, g.seg_dst)
        return self.lin_l(agg, None, res=self.lin_r(x))


class InteractionBlock(nn.Module):
    """pronet.py:150-253."""

    def __init__(self, hidden_channels, output_channels, num_radial, num_spherical, num_layers, mid_emb, act=swish,
                 num_pos_emb=16, dropout=0, level='allatom'):
        super().__init__()
        self.act = act
        self.dropout = nn.Dropout(dropout)
        self.conv0 = EdgeGraphConv(hidden_channels, hidden_channels)
        self.conv1 = EdgeGraphConv(hidden_channels, hidden_channels)
        self.conv2 = EdgeGraphConv(hidden_channels, hidden_channels)
        self.lin_feature0 = TwoLinear(num_radial * num_spherical ** 2, mid_emb, hidden_channels)
        if level == 'aminoacid':
            self.lin_feature1 = TwoLinear(num_radial * num_spherical, mid_emb, hidden_channels)
        elif level in ('backbone', 'allatom'):
            self.lin_feature1 = TwoLinear(3 * num_radial * num_spherical, mid_emb, hidden_channels)
        self.lin_feature2 = TwoLinear(num_pos_emb, mid_emb, hidden_channels)
        self.lin_1 = Linear(hidden_channels, hidden_channels)
        self.lin_2 = Linear(hidden_channels, hidden_channels)
        self.lin0 = Linear(hidden_channels, hidden_channels)
        self.lin1 = Linear(hidden_channels, hidden_channels)
        self.lin2 = Linear(hidden_channels, hidden_channels)
        self.lins_cat = nn.ModuleList([Linear(3 * hidden_channels, hidden_channels)] +
                                      [Linear(hidden_channels, hidden_channels) for _ in range(num_layers - 1)])
        self.lins = nn.ModuleList([Linear(hidden_channels, hidden_channels) for _ in range(num_layers - 1)])
        self.final = Linear(hidden_channels, output_channels)
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.conv0, self.conv1, self.conv2, self.lin_feature0, self.lin_feature1, self.lin_feature2, self.lin_1,
                  self.lin_2, self.lin0, self.lin1, self.lin2, *self.lins, *self.lins_cat, self.final):
            m.reset_parameters()

    def forward(self, x, feature0, feature1, pos_emb, g):
        x_lin_1 = self.lin_1(x, self.act)
        x_lin_2 = self.lin_2(x, self.act)
        hs = []
        for conv, lin, feat, f in ((self.conv0, self.lin0, self.lin_feature0, feature0),
                                   (self.conv1, self.lin1, self.lin_feature1, feature1),
                                   (self.conv2, self.lin2, self.lin_feature2, pos_emb)):
            h = lin(conv(x_lin_1, g, feat(f)), self.act)
            hs.append(self.dropout(h))
        h = torch.cat(hs, 1)
        for lin in self.lins_cat:
            h = lin(h, self.act)
        h = h + x_lin_2
        for lin in self.lins:
            h = lin(h, self.act)
        return self.final(h)


class ProNet(nn.Module):
    r"""ProNet (`"Learning Hierarchical Protein Representations via Complete 3D Graph Networks"`); API of
    method/pronet/pronet.py:276-294."""

    def __init__(self, level='aminoacid', num_blocks=4, hidden_channels=128, out_channels=1, mid_emb=64, num_radial=6,
                 num_spherical=2, cutoff=10.0, max_num_neighbors=32, int_emb_layers=3, out_layers=2, num_pos_emb=16,
                 dropout=0, data_augment_eachlayer=False, euler_noise=False):
        super().__init__()
        if level not in ('aminoacid', 'backbone', 'allatom'):
            raise ValueError(f'No supported model! (level={level!r})')
        self.cutoff, self.max_num_neighbors, self.num_pos_emb = cutoff, max_num_neighbors, num_pos_emb
        self.data_augment_eachlayer, self.euler_noise, self.level = data_augment_eachlayer, euler_noise, level
        self.num_radial, self.num_spherical = num_radial, num_spherical
        self.act = swish
        self.tables = BasisTables(num_spherical, num_radial, 'comenet')      # pronet/features.py == comenet/features.py
        if level == 'aminoacid':
            self.embedding = nn.Embedding(num_aa_type, hidden_channels)
        elif level == 'backbone':
            self.embedding = nn.Linear(num_aa_type + num_bb_embs, hidden_channels)
        else:
            self.embedding = nn.Linear(num_aa_type + num_bb_embs + num_side_chain_embs, hidden_channels)
        self.interaction_blocks = nn.ModuleList([
            InteractionBlock(hidden_channels, hidden_channels, num_radial, num_spherical, int_emb_layers, mid_emb, self.act,
                             num_pos_emb, dropout, level) for _ in range(num_blocks)])
        self.lins_out = nn.ModuleList([Linear(hidden_channels, hidden_channels) for _ in range(out_layers - 1)])
        self.lin_out = Linear(hidden_channels, out_channels)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.reset_parameters()

    def reset_parameters(self):
        self.embedding.reset_parameters()
        for m in self.interaction_blocks:
            m.reset_parameters()
        for lin in self.lins_out:
            lin.reset_parameters()
        self.lin_out.reset_parameters()

    def pos_emb(self, g, num_pos_emb=16):
        """pronet.py:362-372 on the engine's edge list."""
        dev = g.src.device
        freq = torch.exp(torch.arange(0, num_pos_emb, 2, dtype=torch.float32, device=dev)
                         * -(math.log(10000.0) / num_pos_emb))
        half = freq.numel()
        out = torch.empty(g.E, 2 * half, dtype=torch.float32, device=dev)
        call('dig3d_pos_emb', ptr(g.src), ptr(g.dst), g.E, ptr(freq), half, ptr(out), _stream())
        return out

    def geometry(self, pos, pos_n, pos_c, g):
        dev = pos.device
        E = g.E
        f = dict(dtype=torch.float32, device=dev)
        dist, theta, phi, a1 = (torch.empty(E, **f) for _ in range(4))
        lvl = 0 if self.level == 'aminoacid' else 1
        a2 = torch.empty(E, **f) if lvl else None
        a3 = torch.empty(E, **f) if lvl else None
        call('dig3d_pronet_geom', ptr(pos), ptr(pos_n) if lvl else None, ptr(pos_c) if lvl else None, ptr(g.src), ptr(g.dst),
             E, g.N, lvl, ptr(dist), ptr(theta), ptr(phi), ptr(a1), ptr(a2), ptr(a3), _stream())
        return dist, theta, phi, a1, a2, a3

    def forward(self, batch_data):
        z = torch.squeeze(batch_data.x.long())
        pos, batch = batch_data.coords_ca.contiguous(), batch_data.batch
        if self.level == 'aminoacid':
            x = self.embedding(z)
        else:
            feats = [torch.squeeze(F.one_hot(z, num_classes=num_aa_type).float()), batch_data.bb_embs]
            if self.level == 'allatom':
                feats.append(batch_data.side_chain_embs)
            x = self.embedding(torch.cat(feats, dim=1))
        g = build_graph(pos, batch, self.cutoff, max_num_neighbors=self.max_num_neighbors, triplets=False)
        pos_emb = self.pos_emb(g, self.num_pos_emb)
        lvl1 = self.level != 'aminoacid'
        dist, theta, phi, a1, a2, a3 = self.geometry(pos, batch_data.coords_n.contiguous() if lvl1 else None,
                                                     batch_data.coords_c.contiguous() if lvl1 else None, g)
        zeros, norms, pref = self.tables.on(pos.device)
        ns, nr = self.num_spherical, self.num_radial
        bes = ops.bessel_basis(dist, self.cutoff, ns, nr, zeros, norms, 0)
        feature0 = ops.sph_basis(bes, None, theta, phi, ns, nr, pref, 1)           # d_theta_phi_emb
        if lvl1:
            if self.euler_noise:
                noise = torch.clip(torch.empty(3, a1.numel(), device=pos.device).normal_(mean=0.0, std=0.025), min=-0.1,
                                   max=0.1)
                a1, a2, a3 = a1 + noise[0], a2 + noise[1], a3 + noise[2]
            feature1 = torch.cat([ops.sph_basis(bes, None, a, None, ns, nr, pref, 1) for a in (a1, a2, a3)], 1)
        else:
            feature1 = ops.sph_basis(bes, None, a1, None, ns, nr, pref, 1)          # d_angle_emb(dist, tau)
        for block in self.interaction_blocks:
            if self.data_augment_eachlayer:
                x = x + torch.clip(torch.empty(x.shape, device=x.device).normal_(mean=0.0, std=0.025), min=-0.1, max=0.1)
            x = block(x, feature0, feature1, pos_emb, g)
        y = ops.segment_sum(x, g.seg_batch)
        for lin in self.lins_out:
            y = self.dropout(self.relu(lin(y)))
        return self.lin_out(y)

    @property
    def num_params(self):
        return sum(p.numel() for p in self.parameters())
