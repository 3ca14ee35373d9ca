"""Differentiable geometry + basis for ``energy_and_force=True`` (run.py:126: force = -d out / d pos with
``create_graph=True``, then ``loss.backward()`` differentiates THROUGH that gradient).

Everything between ``pos`` and the embeddings must therefore be twice differentiable.  The graph (integer work)
comes from the HIP builder; positions reach the edges through the HIP row-gather / segment-sum pair (linear, closed
under differentiation); every non-linear stage — |vec|, angle, torsion, the Bessel table, the harmonics, dist_emb —
is a Function whose backward and double backward are ONE HIP kernel each (dig_amd/diffops.py, csrc/diffgeom.hip:
derivatives by forward-mode dual numbers in float64, VALUES from the same float32 kernels the energy-only route
uses, so the energies of the two routes are identical).  DimeNet++ at the kernel-supported basis sizes never forms the
angular table (``fused_sbf``: csrc/sbf2.hip); otherwise the table products ``bes[idx_kj] (x) Y`` are torch broadcasting
multiplies.
"""
from ... import diffops, ops


def dime_geometry_differentiable(model, pos, g, fused_sbf=False):
    """(rbf, sbf[, tbf]) for SphereNet / DimeNet++ as twice-differentiable functions of ``pos``.
    ``fused_sbf`` (DimeNet++): -> (rbf, None, bes, angle) — the angular table is never formed, the caller contracts
    (bes, angle) with the first basis Linears of all blocks in the basis kernel itself (diffops.sbf_project)."""
    emb = model.emb
    ns, nr = emb.ns, emb.nr
    zeros, norms, pref = emb.tables.on(pos.device)
    posd = pos.detach().contiguous()
    vec = diffops.edge_vectors(pos, g)                                          # pos_i - pos_j  [E,3]
    dist = diffops.edge_len(vec, 0, g.cnt_E)
    if emb.torsion:
        angle, tor = diffops.triplet_angles(vec, posd, g, True)
    else:
        angle, tor = diffops.triplet_angles(vec, posd, g, False), None
    rbf = emb.dist_emb(dist, g.cnt_E)
    bes = diffops.bessel_basis(dist, emb.cutoff, ns, nr, zeros, norms, emb.env_p, g.cnt_E)     # [E, ns*nr]
    if fused_sbf and not emb.torsion:
        return rbf, None, bes, angle
    bes_t = ops.gather_rows(bes, g.seg_kj)                                       # rbf[idx_kj]  [T, ns*nr]
    yl0 = diffops.harmonics(angle, None, ns, pref, g.cnt_T)                      # [T, ns]
    sbf = (bes_t.view(-1, ns, nr) * yl0.unsqueeze(-1)).reshape(-1, ns * nr)
    if not emb.torsion:
        return rbf, sbf
    ylm = diffops.harmonics(angle, tor, ns, pref, g.cnt_T)                       # [T, ns*ns]
    # spherenet/features.py:262: cbf.view(-1, ns, ns, 1) * rbf.view(-1, 1, ns, nr)
    tbf = (bes_t.view(-1, 1, ns, nr) * ylm.view(-1, ns, ns, 1)).reshape(-1, ns * ns * nr)
    return rbf, sbf, tbf
