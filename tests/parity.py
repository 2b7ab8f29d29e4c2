"""Parity assertions shared by the GPU tests (BASELINE.json north_star: "matrix entries matching to 1e-12", fp64).

Two bounds, both against the oracle on identical inputs:
  * global:     max|A_gpu - A_ref| <= tol * max|A_ref|
  * per block:  for every n x n block (P, Q) of the pattern
                max|block_gpu - block_ref| <= tol * max(max|block_ref|, floor * max|A_ref|)
    - a coupling block is 10..1000 times smaller than the diagonal blocks, so the global bound alone would let it lose
    digits unnoticed; `floor` keeps blocks that vanish identically (to rounding) from asking for relative accuracy of zero.
The unit of the relative bound is the block, not the entry: an entry is a sum of terms of the size of its block's largest
entries, and that is what rounding is relative to.
"""
import numpy as np

TOL = 1e-12
FLOOR = 1e-6


def block_keys(rowptr, colind, n):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    rows = np.repeat(np.arange(len(rowptr) - 1, dtype=np.int64), np.diff(rowptr))
    nbc = int(colind.max()) // n + 1 if len(colind) else 1
    return (rows // n) * nbc + colind // n


def assert_parity(got, ref, rowptr, colind, n, tol=TOL, floor=FLOOR, what=""):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.all(np.isfinite(got)), what
    scale = float(np.max(np.abs(ref))) if ref.size else 0.0
    err = np.abs(got - ref)
    gmax = float(err.max()) if err.size else 0.0
    assert gmax <= tol * scale, "%s global: max err %.3e = %.3e * max|A|" % (what, gmax, gmax / max(scale, 1e-300))
    keys = block_keys(rowptr, colind, n)
    assert keys.shape == ref.shape
    _, inv = np.unique(keys, return_inverse=True)
    nb = int(inv.max()) + 1 if inv.size else 0
    bmax = np.zeros(nb)
    berr = np.zeros(nb)
    np.maximum.at(bmax, inv, np.abs(ref))
    np.maximum.at(berr, inv, err)
    bound = tol * np.maximum(bmax, floor * scale)
    bad = np.nonzero(berr > bound)[0]
    assert bad.size == 0, "%s per block: %d of %d blocks off, worst err/bound %.3e (block max %.3e, global max %.3e)" % (
        what, bad.size, nb, float((berr / bound).max()), float(bmax[np.argmax(berr / bound)]), scale)
    return gmax / max(scale, 1e-300)


def assert_parity_ah(got, ref, ah, diag_first=True, rows=None, **kw):
    """Same, with the pattern taken from an oracle handler; rows = (r0, r1): `got` / `ref` hold the values of that row
    range only (rank-local assemblies)."""
    cache = ah.__dict__.setdefault("_sp_cache", {})
    if diag_first not in cache:
        cache[diag_first] = ah.sparsity_pattern(diag_first)
    rp, ci = cache[diag_first]
    n = ah.fe.n_dofs_per_cell
    if rows is not None:
        r0, r1 = rows
        ci = ci[rp[r0]:rp[r1]]
        rp = rp[r0:r1 + 1] - rp[r0]
    return assert_parity(got, ref, rp, ci, n, **kw)
