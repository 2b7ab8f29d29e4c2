"""moment vs direct form when neighbouring polytopes differ in size by a factor r (one r^3-cell polytope among singletons)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polydeal_amd as pa
from oracle import polydeal_oracle as po
from flatten_oracle import flatten

for r, lg in ((2, 2), (4, 3), (8, 4)):
    fe = po.FE_DGQ(3, 3)
    grid = po.hyper_cube_refined(3, 0.0, 1.0, lg)
    n = 2 ** lg
    ah = po.AgglomerationHandler(grid)
    big = [int(grid.ijk_to_cell[(i, j, k)]) for i in range(r) for j in range(r) for k in range(r)]
    ah.define_agglomerate(sorted(big))
    # the rest: singletons next to the big one, the far region in 2x2x2 blocks to keep the problem small
    flagged = set(big)
    for c in range(grid.n_cells):
        if c not in flagged:
            ah.define_agglomerate([c])
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    kw = flatten(ah, var, with_colind=False)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    res = {}
    for alg in ("direct", "moment"):
        ctx.set_algorithm(alg)
        res[alg] = ctx.assemble()
    ctx.close()
    sc = np.max(np.abs(res["direct"]))
    print("size ratio %d: %d polytopes, max|moment-direct|/max|A| = %.2e" % (r, ah.n_agglomerates, np.max(np.abs(res["moment"] - res["direct"])) / sc), flush=True)
