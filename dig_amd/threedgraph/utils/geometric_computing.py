"""``xyz_to_dat`` drop-in (dig/threedgraph/utils/geometric_computing.py:12-80) on the HIP engine."""
import torch

from ... import ops
from ...graph import graph_from_edge_index


def xyz_to_dat(pos, edge_index, num_nodes, use_torsion=False):
    """Distance, angle (and torsion) plus the triplet index lists.

    Returns ``(dist, angle, i, j, idx_kj, idx_ji)`` or, with ``use_torsion``,
    ``(dist, angle, torsion, i, j, idx_kj, idx_ji)`` — same order and dtypes (float32 / int64) as the
    reference.  ``dist`` / ``angle`` / ``torsion`` are differentiable w.r.t. ``pos`` to second order (the reference
    model calls ``pos.requires_grad_()`` and trains forces through this function: spherenet.py:302-306, run.py:126):
    same values as the forward-only kernels, derivative kernels from csrc/diffgeom.hip."""
    j, i = edge_index
    g = graph_from_edge_index(edge_index, num_nodes, triplets=True)
    posc = pos.detach().contiguous()
    if pos.requires_grad and torch.is_grad_enabled():
        from ... import diffops
        vec = diffops.edge_vectors(pos, g)
        dist = diffops.edge_len(vec, 0, None)
        if use_torsion:
            angle, torsion = diffops.triplet_angles(vec, posc, g, True)
        else:
            angle, torsion = diffops.triplet_angles(vec, posc, g, False), None
    else:
        dist = ops.edge_dist(posc, g, 0)
        angle, torsion, _ = ops.triplet_geom(posc, g, use_torsion)
    idx_kj, idx_ji = g.idx_kj_ji
    if use_torsion:
        return dist, angle, torsion, i, j, idx_kj, idx_ji
    return dist, angle, i, j, idx_kj, idx_ji
