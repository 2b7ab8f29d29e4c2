"""Post-processing on the HIP path: evaluation of the polytopal solution AND the weighted error sums on
the device (pdh_global_error: one kernel, 16 bytes per polytope come back).

Mirrors PolyUtils::compute_global_error (reference include/poly_utils.h:1647-1750) and the evaluation step of
PolyUtils::interpolate_to_fine_grid (:1145-1274).  The evaluation itself is ``Context.evaluate`` ->
``pdh_evaluate`` (csrc/pdh_eval.hip); there is no CPU evaluation path here.
"""
from __future__ import annotations

import numpy as np

from ._capi import Context


def compute_global_error(ctx: Context, vq_ptr, vq_x, vq_w, solution, exact, exact_grad=None):
    """(L2 error, H1-seminorm error or None) of `solution` on the problem resident in `ctx`.

    vq_ptr/vq_x/vq_w are the volume-quadrature arrays the problem was set with; `exact(x[N,dim]) -> [N]`,
    `exact_grad(x) -> [N,dim]`.  Only the polytopes owned by `ctx` contribute (sum the squares over ranks for a
    partitioned problem, as the reference does with Utilities::MPI::sum, poly_utils.h:1736-1745)."""
    vq_x = np.asarray(vq_x, dtype=np.float64)
    vq_w = np.asarray(vq_w, dtype=np.float64)
    xs = vq_x.T
    eu = np.asarray(exact(xs), dtype=np.float64)
    eg = np.zeros_like(vq_x) if exact_grad is None else np.ascontiguousarray(np.asarray(exact_grad(xs), dtype=np.float64).T)
    l2, h1 = ctx.global_error_sums(solution, vq_ptr, vq_x, vq_w, eu, eg)
    return np.sqrt(l2), (None if exact_grad is None else np.sqrt(h1))


def interpolate_to_points(ctx: Context, solution, pt_ptr, pts):
    """u_h at caller-given points per polytope - with the support points of the sub-cells this is the vector
    PolyUtils::interpolate_to_fine_grid writes (include/poly_utils.h:1196-1233)."""
    return ctx.evaluate(solution, pt_ptr, pts)
