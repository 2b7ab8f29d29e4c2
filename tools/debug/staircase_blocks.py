#!/usr/bin/env python3
"""Diagnostics: which n x n blocks of the staircase case differ from the oracle, and by how much (per polytope pair)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import polydeal_amd as pa  # noqa: E402
from flatten_oracle import flatten  # noqa: E402
from oracle import polydeal_oracle as po  # noqa: E402
from test_gpu_parity import _staircase_groups  # noqa: E402

fe = po.FE_DGQ(3, 3)
grid = po.hyper_cube_refined(3, 0.0, 1.0, 2)
ah = po.AgglomerationHandler(grid)
for g in _staircase_groups(grid):
    ah.define_agglomerate(g)
ah.initialize_fe_values(4, 4)
ah.distribute_agglomerated_dofs(fe)
var = po.variant_poisson_example(fe)
kw = flatten(ah, var, diag_first=False)
rp, ci, ref = po.assemble_csr(ah, var, diag_first=False)
ctx = pa.Context(0)
ctx.set_algorithm(sys.argv[1] if len(sys.argv) > 1 else "auto")
ctx.set_problem(pa.Problem(**kw))
print("algorithm", ctx.algorithm_in_use())
v = ctx.assemble()
n = fe.n_dofs_per_cell
N = ah.n_dofs
A = sp.csr_matrix((v, ci, rp), shape=(N, N)).toarray()
R = sp.csr_matrix((ref, ci, rp), shape=(N, N)).toarray()
nA = N // n
sc = np.max(np.abs(R))
for P in range(nA):
    for Q in range(nA):
        a, r = A[P * n:(P + 1) * n, Q * n:(Q + 1) * n], R[P * n:(P + 1) * n, Q * n:(Q + 1) * n]
        if np.max(np.abs(r)) == 0 and np.max(np.abs(a)) == 0:
            continue
        e = np.max(np.abs(a - r)) / sc
        if e > 1e-12:
            i, j = np.unravel_index(np.argmax(np.abs(a - r)), a.shape)
            print("block (%d,%d): err %.3e at (%d,%d) got %.6e want %.6e; ratio of norms %.6f" % (P, Q, e, i, j, a[i, j], r[i, j], np.linalg.norm(a) / max(np.linalg.norm(r), 1e-300)))
ctx.close()
