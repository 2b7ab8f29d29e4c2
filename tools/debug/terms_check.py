#!/usr/bin/env python3
"""Development check of the term kernel (pdh_terms.h) against the oracle: block and staircase agglomerates, all five small
elements, both CSR layouts, every variant; prints the error per case (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import polydeal_amd as pa  # noqa: E402
from flatten_oracle import flatten  # noqa: E402
from test_gpu_parity import random_agglomeration  # noqa: E402
from oracle import polydeal_oracle as po  # noqa: E402
from parity import assert_parity_ah  # noqa: E402


def values(kw, alg="auto", r0=0, r1=None):
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_algorithm(alg)
    ctx.set_problem(prob, r0, r1)
    used = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
    v = ctx.assemble()
    ctx.close()
    return v, used


def variant(name, fe):
    return {"test": po.variant_minimal_sip_test, "adm": po.variant_assemble_dg_matrix, "poisson": lambda: po.variant_poisson_example(fe),
            "dr": lambda: po.variant_diffusion_reaction(fe), "minsip": po.variant_minimal_sip_example}[name]()


bad = 0
for basis, p in (("dgp", 3), ("dgq", 2), ("dgp", 2), ("dgq", 1), ("dgp", 1)):
    for mode, cells, per, vname, diag_first in (("block", 4, 2, "poisson", True), ("block", 4, 2, "dr", False), ("block", 2, 2, "adm", True),
                                                ("block", 4, 1, "test", True), ("grown", 4, 4, "poisson", True), ("grown", 6, 6, "dr", False),
                                                ("grown", 6, 3, "minsip", True), ("block", 6, 3, "adm", False)):
        fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
        grid = po.subdivided_hyper_cube(3, cells, 0.0, 1.0)
        ah = po.AgglomerationHandler(grid)
        if mode == "block":
            groups = po.block_agglomerates(grid, per)
        else:
            groups = random_agglomeration(grid, max(2, grid.n_cells // per), np.random.default_rng(cells + per))
        for g in groups:
            ah.define_agglomerate(g)
        nq = p + 1
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        var = variant(vname, fe)
        kw = flatten(ah, var, diag_first=diag_first)
        ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
        v, used = values(kw)
        err = np.max(np.abs(v - ref)) / np.max(np.abs(ref))
        ok = True
        try:
            assert_parity_ah(v, ref, ah, diag_first, what="terms")
        except AssertionError as e:
            ok = False
            bad += 1
            print("   ", str(e)[:300])
        print("%s p=%d %-5s cells=%d per=%d %-7s diag_first=%d: %s err %.2e %s" % (basis, p, mode, cells, per, vname, diag_first, used, err,
                                                                                  "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
