"""Work-balanced row partition (polydeal_amd/partition.py): contiguous ranges of whole polytopes that balance the non-zeros a rank
writes.  The reference partitions the cell graph by work (include/poly_utils.h:553-704); with METIS-like agglomerates the number of
neighbours per polytope varies 4x, and equal polytope counts leave the ranks 10-20 % apart."""
import ctypes as C

import numpy as np
import pytest

import polydeal_amd as pa
from polydeal_amd.partition import balanced_row_splits, row_range


def _grown(cells=12, per=8, seed=1, basis="dgp", p=2):
    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    ah.define_grown_agglomerates(per, seed=seed)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(3, p)
    ah.initialize_fe_values(p + 1, p + 1)
    ah.distribute_agglomerated_dofs(fe)
    return ah, fe


def test_blocks_per_row_are_the_row_lengths_of_the_pattern():
    ah, fe = _grown()
    n = fe.n_dofs_per_cell
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    rp = flat.arrays()["rowptr"]
    w = ah.blocks_per_row()
    assert np.array_equal(np.diff(rp)[::n], w.astype(np.int64) * n)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_balanced_splits_beat_equal_counts_on_grown_agglomerates(world):
    ah, fe = _grown()
    n = fe.n_dofs_per_cell
    w = ah.blocks_per_row()
    sp = balanced_row_splits(w, n, world)
    assert sp[0] == 0 and sp[-1] == ah.n_dofs and all(b > a for a, b in zip(sp, sp[1:])) and all(s % n == 0 for s in sp)
    bal = [int(w[sp[r] // n:sp[r + 1] // n].sum()) for r in range(world)]
    cnt = [int(w[a // n:b // n].sum()) for a, b in (row_range(len(w), n, r, world) for r in range(world))]
    assert max(bal) / min(bal) <= max(cnt) / min(cnt) + 1e-12
    assert max(bal) - min(bal) <= 2 * int(w.max())  # within a polytope or two of each other
    # every range is a valid owned range for the library (whole polytopes; host-only check, no GPU)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    lib = pa.load_library()
    for r in range(world):
        stats = (C.c_int64 * 8)()
        assert lib.pdh_check_problem(C.byref(flat.c), sp[r], sp[r + 1], stats) == 0
        assert stats[0] == (sp[r + 1] - sp[r]) // n and stats[4] == bal[r] * n * n


def test_balanced_splits_degenerate_cases():
    assert balanced_row_splits([3, 3, 3, 3], 5, 1) == [0, 20]
    assert balanced_row_splits([1, 1, 1, 1], 2, 4) == [0, 2, 4, 6, 8]
    assert balanced_row_splits([10, 1, 1, 1, 1], 1, 2) == [0, 1, 5]
    sp = balanced_row_splits([2, 2], 3, 4)  # fewer polytopes than ranks: empty ranges at the end, still monotone
    assert sp[0] == 0 and sp[-1] == 6 and all(b >= a for a, b in zip(sp, sp[1:]))
    with pytest.raises(ValueError):
        balanced_row_splits([1], 1, 0)
