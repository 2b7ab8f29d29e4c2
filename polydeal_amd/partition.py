"""Row partition of the polytopes over ranks (multi-GPU): contiguous dof-row ranges of whole polytopes,
like deal.II's locally_owned_dofs for whole agglomerates per rank (reference
source/agglomeration_handler.cc:83-87, examples/diffusion_reaction.cc:448)."""


def polytope_range(n_agg: int, rank: int, world: int):
    """Polytopes (in dof order) owned by `rank`: [a0, a1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n_agg * rank) // world, (n_agg * (rank + 1)) // world


def row_range(n_agg: int, dofs_per_cell: int, rank: int, world: int):
    a0, a1 = polytope_range(n_agg, rank, world)
    return a0 * dofs_per_cell, a1 * dofs_per_cell


def balanced_row_splits(blocks_per_row, dofs_per_cell: int, world: int):
    """First dof row of every rank (+ the total) for contiguous ranges of whole polytopes that balance the NON-ZEROS a rank writes:
    a polytope's rows hold n * n * (1 + neighbours) values, and on METIS-like agglomerates the number of neighbours varies from 4
    to 40 - equal polytope COUNTS then leave ranks 10-20 % apart in bytes to write.  blocks_per_row: 1 + neighbours of every
    polytope in dof order (AgglomerationHandler.blocks_per_row()).  Cut k is placed where the running sum of weights is nearest to
    k / world of the total; every rank gets at least one polytope while there are enough."""
    import numpy as np

    w = np.asarray(blocks_per_row, dtype=np.float64)
    n_agg = len(w)
    if world < 1:
        raise ValueError("world must be >= 1")
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, world):
        target = cum[-1] * k / world
        a = int(np.searchsorted(cum, target))  # first prefix that reaches the target ...
        if a > 0 and abs(cum[a - 1] - target) <= abs(cum[min(a, n_agg)] - target):
            a -= 1                             # ... or the one before it, whichever is nearer
        lo, hi = cuts[-1] + 1, n_agg - (world - k)  # at least one polytope for this rank and for every rank still to come
        a = min(max(a, lo), hi) if hi >= lo else min(max(a, cuts[-1]), n_agg)
        cuts.append(a)
    cuts.append(n_agg)
    return [c * dofs_per_cell for c in cuts]
