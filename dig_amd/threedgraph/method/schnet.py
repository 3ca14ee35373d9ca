"""SchNet on the HIP engine.  Drop-in for ``dig.threedgraph.method.SchNet``
(method/schnet/schnet.py:106-168): same constructor keywords/defaults, ``forward(batch_data)``, and
``state_dict`` layout (init_v.weight, dist_emb.offset, update_es.{i}.lin/mlp.0/mlp.2, update_vs.{i}.lin1/lin2,
update_u.lin1/lin2).

The continuous-filter convolution ``e = lin(v)[j] * W ; scatter(e, i)`` (schnet.py:29-35,53-55) is ONE fused
HIP segment kernel (gather by source, multiply by the filter, sum per target), forward and backward.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ..._hip import call, ptr
from ...graph import build_graph, _stream
from ..data import check_z_bounds

_LOG2_F32 = torch.log(torch.tensor(2.0)).item()      # ShiftedSoftplus shift (schnet.py:100)


def shifted_softplus(x):
    return F.softplus(x) - _LOG2_F32


class _SSP(nn.Module):
    def forward(self, x):
        return shifted_softplus(x)


class _Filter(nn.Module):
    """``update_e`` (schnet.py:9-35): W = mlp(gauss(d)) * cosine_cutoff(d);  messages v_lin[j] * W."""

    def __init__(self, hidden, filters, gaussians, cutoff):
        super().__init__()
        self.cutoff = cutoff
        self.lin = nn.Linear(hidden, filters, bias=False)
        self.mlp = nn.Sequential(nn.Linear(gaussians, filters), _SSP(), nn.Linear(filters, filters))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin.weight)
        nn.init.xavier_uniform_(self.mlp[0].weight)
        self.mlp[0].bias.data.zero_()
        nn.init.xavier_uniform_(self.mlp[2].weight)   # mlp[2].bias keeps its default init (schnet.py:25-27)

    def forward(self, v, dist_emb, C, w=None, W=None):
        # lin (no bias), mlp = Linear -> ssp -> Linear: f32-MFMA kernels (csrc/dense.hip)
        # (w: mlp[0] + ssp of this block, already evaluated with those of the other blocks in one grouped launch)
        if w is None:
            w = ops.linear(dist_emb, self.mlp[0].weight, self.mlp[0].bias, ops.ACT_SSP)
        v_lin, v = ops.linear_tap(v, self.lin.weight)       # v' = alias of v for the residual of update_v (schnet.py:59)
        if W is None:          # (else: the whole filter network of every block was evaluated in two grouped launches)
            W = ops.linear_rowscale(w, self.mlp[2].weight, self.mlp[2].bias, C)
        return v_lin, W, v


class _NodeUpdate(nn.Module):
    """``update_v`` (schnet.py:38-59)."""

    def __init__(self, hidden, filters):
        super().__init__()
        self.lin1 = nn.Linear(filters, hidden)
        self.lin2 = nn.Linear(hidden, hidden)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.lin1, self.lin2):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()

    def forward(self, v, agg):
        h = ops.linear(agg, self.lin1.weight, self.lin1.bias, ops.ACT_SSP)
        return ops.linear(h, self.lin2.weight, self.lin2.bias, ops.ACT_NONE, res=v)


class _Readout(nn.Module):
    """``update_u`` (schnet.py:62-82)."""

    def __init__(self, hidden, out_channels):
        super().__init__()
        self.lin1 = nn.Linear(hidden, hidden // 2)
        self.lin2 = nn.Linear(hidden // 2, out_channels)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.lin1, self.lin2):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()

    def forward(self, v, g):
        h = ops.linear(v, self.lin1.weight, self.lin1.bias, ops.ACT_SSP)
        return ops.segment_sum(ops.linear(h, self.lin2.weight, self.lin2.bias), g.seg_batch)


class _Gaussians(nn.Module):
    """``emb`` (schnet.py:85-94)."""

    def __init__(self, start, stop, num_gaussians):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer('offset', offset)

    def forward(self, dist):
        if dist.requires_grad:
            d = dist.view(-1, 1) - self.offset.view(1, -1)
            return torch.exp(self.coeff * d * d)
        E, G = dist.numel(), self.offset.numel()
        out = torch.empty(E, G, dtype=torch.float32, device=dist.device)
        call('dig3d_gauss_smear', ptr(dist), E, ptr(self.offset), G, float(self.coeff), ptr(out), _stream())
        return out


class SchNet(nn.Module):
    r"""SchNet re-implementation under the 3DGN framework; API of method/schnet/schnet.py:120."""

    def __init__(self, energy_and_force=False, cutoff=10.0, num_layers=6, hidden_channels=128, out_channels=1,
                 num_filters=128, num_gaussians=50):
        super().__init__()
        self.energy_and_force = energy_and_force
        self.cutoff = cutoff
        self.num_layers = num_layers
        self.hidden_channels = hidden_channels
        self.out_channels = out_channels
        self.num_filters = num_filters
        self.num_gaussians = num_gaussians
        self.init_v = nn.Embedding(100, hidden_channels)
        self.dist_emb = _Gaussians(0.0, cutoff, num_gaussians)
        self.update_vs = nn.ModuleList([_NodeUpdate(hidden_channels, num_filters) for _ in range(num_layers)])
        self.update_es = nn.ModuleList([_Filter(hidden_channels, num_filters, num_gaussians, cutoff)
                                        for _ in range(num_layers)])
        self.update_u = _Readout(hidden_channels, out_channels)
        self.reset_parameters()

    def reset_parameters(self):
        self.init_v.reset_parameters()
        for m in self.update_es:
            m.reset_parameters()
        for m in self.update_vs:
            m.reset_parameters()
        self.update_u.reset_parameters()

    needs_triplets = False          # dig_amd/graphed.py: the radius graph without triplet lists

    def _fused_ok(self):
        return True

    def forward(self, batch_data):
        check_z_bounds(batch_data, self.init_v.num_embeddings)
        if getattr(batch_data, 'is_static_graph', False):      # dig_amd/graphed.py: padded, prebuilt graph
            pos = batch_data.pos_leaf if self.energy_and_force else batch_data.pos
            with ops.composite_mode(self.energy_and_force):
                return self._forward(batch_data.z, pos, None, batch_data)
        z, pos, batch = batch_data.z, batch_data.pos, batch_data.batch
        if self.energy_and_force:
            pos.requires_grad_()
        with ops.composite_mode(pos.requires_grad):    # forces: twice-differentiable route
            return self._forward(z, pos, batch)

    def _forward(self, z, pos, batch, g=None):
        if g is None:
            g = build_graph(pos, batch, self.cutoff, triplets=False)
        if pos.requires_grad:
            # differentiable distances: HIP row gathers + |vec| with its first and second derivative kernels
            from ... import diffops
            dist = diffops.edge_len(ops.gather_rows(pos, g.seg_src) - ops.gather_rows(pos, g.seg_dst), 1, g.cnt_E)
            C = 0.5 * (torch.cos(dist * math.pi / self.cutoff) + 1.0)
        else:
            dist = ops.edge_dist(pos.contiguous(), g, 1)
            C = torch.empty_like(dist)
            call('dig3d_cos_cutoff', ptr(dist), g.E, float(self.cutoff), ptr(C), _stream())
        dist_emb = self.dist_emb(dist)
        v = ops.embedding(z, self.init_v.weight)
        # the first filter-generating layer of EVERY block reads the same Gaussian rows and nothing of the node states
        # (schnet.py:29-31): one grouped launch per pass instead of one per block
        ws = None
        first = [m.mlp[0] for m in self.update_es]
        if (ops.schnet_group_filters and not pos.requires_grad and 1 < len(first) <= 8
                and ops.grouped_linear_supported([dist_emb] * len(first), [l.weight for l in first])):
            ws = ops.grouped_linear([dist_emb] * len(first), [l.weight for l in first], [l.bias for l in first], ops.ACT_SSP)
        # ... and so does the second one (its input is the first one's output; the cosine cutoff is a row factor)
        Wf = None
        second = [m.mlp[2] for m in self.update_es]
        if (ws is not None and ws[0].size(1) > 16 and not C.requires_grad and C.dtype == torch.float32
                and all(l.bias is not None for l in second) and ops.grouped_linear_supported(ws, [l.weight for l in second])):
            Wf = ops.grouped_linear_rowscale(ws, [l.weight for l in second], [l.bias for l in second], C)
        for l, (upd_e, upd_v) in enumerate(zip(self.update_es, self.update_vs)):
            v_lin, W, v = upd_e(v, dist_emb, C, ws[l] if ws is not None else None, Wf[l] if Wf is not None else None)
            if pos.requires_grad:
                agg = ops.segment_sum(ops.gather_rows(v_lin, g.seg_src) * W, g.seg_dst)
            else:
                agg = ops.gather_mul_segment_sum(v_lin, W, None, g.seg_src, g.seg_dst)
            v = upd_v(v, agg)
        return self.update_u(v, g)
