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
