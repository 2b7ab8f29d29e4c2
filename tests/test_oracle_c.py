"""The plain-C restatement (oracle/sip_ref.c, also the timed CPU baseline) vs the NumPy oracle."""
import numpy as np
import pytest

from flatten_oracle import flatten
from oracle import polydeal_oracle as po
from oracle import sip_ref


@pytest.mark.parametrize("dim,lg,b,fe_cls,p,nq,var,dist", [
    (2, 3, 2, po.FE_DGQ, 2, 3, "adm", 0.0),
    (2, 3, 2, po.FE_AggloDGP, 3, 4, "poisson", 0.2),
    (3, 2, 2, po.FE_DGQ, 1, 2, "test", 0.0),
    (3, 2, 2, po.FE_AggloDGP, 2, 3, "dr", 0.1),
    (3, 1, 1, po.FE_DGQ, 3, 4, "poisson", 0.0),
])
def test_c_restatement_matches_numpy_oracle(dim, lg, b, fe_cls, p, nq, var, dist):
    grid = po.hyper_cube_refined(dim, 0.0, 1.0, lg)
    if dist:
        grid.distort(dist, seed=2)
    ah = po.AgglomerationHandler(grid)
    for g in po.block_agglomerates(grid, b):
        ah.define_agglomerate(g)
    fe = fe_cls(dim, p)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    v = {"adm": po.variant_assemble_dg_matrix, "test": po.variant_minimal_sip_test,
         "poisson": lambda: po.variant_poisson_example(fe), "dr": lambda: po.variant_diffusion_reaction(fe)}[var]()
    _, _, ref = po.assemble_csr(ah, v)
    kw = flatten(ah, v)
    got, _ = sip_ref.assemble(kw, nthreads=1)
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))
    got2, _ = sip_ref.assemble(kw, nthreads=2)
    assert np.max(np.abs(got2 - ref)) <= 1e-13 * np.max(np.abs(ref))
    # hoisted / vectorised CPU variant (second CPU baseline of bench.py)
    for thr in (1, 2):
        got3, _ = sip_ref.assemble(kw, nthreads=thr, fast=True)
        assert np.max(np.abs(got3 - ref)) <= 1e-13 * np.max(np.abs(ref))
