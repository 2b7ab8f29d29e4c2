"""One-off fuzz: moment vs direct form on random irregular agglomerations (3-D, all elements with a moment form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polydeal_amd as pa
from oracle import polydeal_oracle as po
from flatten_oracle import flatten
from test_gpu_parity import random_agglomeration

def vals(kw, alg):
    prob = pa.Problem(**kw); ctx = pa.Context(0); ctx.set_algorithm(alg); ctx.set_problem(prob)
    v = ctx.assemble(); ctx.close(); return v

worst = 0.0
for seed in range(24):
    rng = np.random.default_rng(100 + seed)
    fe_cls, p = [(po.FE_DGQ, 3), (po.FE_DGQ, 2), (po.FE_DGQ, 1), (po.FE_AggloDGP, 3), (po.FE_AggloDGP, 2), (po.FE_AggloDGP, 1)][seed % 6]
    fe = fe_cls(3, p)
    lg = 2 if seed % 3 else 3
    grid = po.hyper_cube_refined(3, -1.0, 2.0, lg)
    grid.distort(0.05 + 0.2 * rng.random(), seed=seed)
    ah = po.AgglomerationHandler(grid)
    nseeds = int(rng.integers(2, 12 if lg == 2 else 40))
    for g in random_agglomeration(grid, nseeds, rng, allow_disconnected=bool(seed % 2)):
        ah.define_agglomerate(g)
    nq = p + 1 + int(rng.integers(0, 2))
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = [po.variant_poisson_example(fe), po.variant_diffusion_reaction(fe), po.variant_assemble_dg_matrix(), po.variant_minimal_sip_example()][seed % 4]
    kw = flatten(ah, var, diag_first=bool(seed % 5), with_colind=False)
    d = vals(kw, "direct"); m = vals(kw, "moment")
    err = np.max(np.abs(m - d)) / np.max(np.abs(d))
    worst = max(worst, err)
    print("seed %2d %s(%d) lg=%d polytopes=%3d nq=%d var=%d: %.2e%s" % (seed, fe.name, p, lg, ah.n_agglomerates, nq, seed % 4, err, "  <-- " if err > 1e-12 or not np.isfinite(err) else ""), flush=True)
print("worst", worst)
