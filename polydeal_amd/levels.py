"""Per-level SIP assembly for agglomerated multigrid hierarchies.

The reference builds one AgglomerationHandler per R-tree level and assembles the level operator with the same
hot path (examples/simplex_agglomerated_multigrid.cc:378-390 calls PolyUtils::assemble_dg_matrix once per
level; include/poly_utils.h:1761-1862 builds the level handlers).  On structured grids the R-tree levels are
exactly blocks of 2^k cells per direction (test/polydeal/rtree_mesh.output, 3DRtree.output), which is what
`block_hierarchy` produces; each level is assembled by the same HIP kernels."""
from __future__ import annotations

from .handler import AgglomerationHandler, BackgroundGrid, FiniteElement, SipVariant, assemble_dg_matrix  # noqa: F401


def block_hierarchy(grid: BackgroundGrid, fe: FiniteElement, blocks, n_q_points_1d=None):
    """One handler per level; `blocks` = cells per direction of a polytope on each level, coarse to fine
    (e.g. [8, 4, 2] on a 64^3 grid gives 8^3, 16^3, 32^3 polytopes)."""
    nq = n_q_points_1d or fe.degree + 1
    levels = []
    for b in blocks:
        ah = AgglomerationHandler(grid)
        ah.define_block_agglomerates(b)
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        levels.append(ah)
    return levels


def assemble_levels(levels, fe: FiniteElement, variant: SipVariant | None = None, diag_first=True, device=0):
    """[(rowptr, colind, values)] - the 'V-cycle assembly' of BASELINE.json configs[4]: one assemble_dg_matrix per level
    (examples/simplex_agglomerated_multigrid.cc:378-390), all on ONE context (stream, events, scratch buffers are created once;
    every level replaces the resident problem).  Levels of Cartesian cells are handed over without their quadrature points
    (flatten_cartesian: generated on the device) where the term kernels take them; distorted cells and polytopes too large for those
    kernels go through the points-based description as before."""
    from ._capi import Context, PdhError
    from .handler import HostError

    variant = variant or SipVariant.assemble_dg_matrix()
    out = []
    ctx = Context(device)
    try:
        for ah in levels:
            if fe != ah.fe:
                raise ValueError("FE passed to assemble_levels differs from a level handler's")
            flat = None
            if ah.grid.dim == 3:
                try:
                    flat = ah.flatten_cartesian(variant, diag_first, True)
                    ctx.set_problem(flat)
                except (PdhError, HostError):
                    flat = None
            if flat is None:
                flat = ah.flatten(variant, diag_first, True)
                ctx.set_problem(flat)
            arr = flat.arrays()
            out.append((arr["rowptr"].copy(), arr["colind"].copy(), ctx.assemble()))
    finally:
        ctx.close()
    return out
