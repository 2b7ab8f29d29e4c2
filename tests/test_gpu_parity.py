"""GPU parity: HIP path (through the C ABI) vs the oracle on identical flattened inputs."""
import os

import numpy as np
import pytest

import golden_cases as gc
from flatten_oracle import flatten
from oracle import polydeal_oracle as po
from parity import assert_parity, assert_parity_ah

pytestmark = pytest.mark.gpu

# north_star: entries within 1e-12 relative (fp64).  Matrix comparisons go through tests/parity.py: 1e-12 of max|A| globally
# AND 1e-12 of every n x n block's own largest entry (floor 1e-6 max|A|); TOL alone is used for vectors / scalars.
TOL = 1e-12


def gpu_values(kw):
    import polydeal_amd as pa

    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    v = ctx.assemble()
    ctx.close()
    return v


def build(dim, n_per_dir_log2, b, fe, nq, distort=0.0, morton=True, lo=0.0, hi=1.0):
    grid = po.hyper_cube_refined(dim, lo, hi, n_per_dir_log2) if morton else po.subdivided_hyper_cube(dim, 2 ** n_per_dir_log2, lo, hi)
    if distort:
        grid.distort(distort, seed=3)
    ah = po.AgglomerationHandler(grid)
    for g in po.block_agglomerates(grid, b):
        ah.define_agglomerate(g)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    return ah


CASES = [
    # (dim, log2 cells/dir, block, fe ctor, degree, nq, variant name, distort)
    (2, 3, 2, po.FE_DGQ, 1, 3, "test", 0.0),
    (2, 3, 2, po.FE_DGQ, 2, 3, "adm", 0.0),
    (2, 3, 2, po.FE_AggloDGP, 2, 3, "poisson", 0.2),
    (2, 2, 2, po.FE_DGQ, 3, 4, "adm", 0.0),
    (2, 3, 4, po.FE_AggloDGP, 3, 4, "poisson", 0.0),
    (3, 2, 2, po.FE_DGQ, 1, 2, "test", 0.0),
    (3, 2, 2, po.FE_AggloDGP, 1, 2, "poisson", 0.0),
    (3, 2, 2, po.FE_AggloDGP, 2, 3, "poisson", 0.15),
    (3, 2, 2, po.FE_DGQ, 2, 3, "dr", 0.0),
    (3, 2, 2, po.FE_AggloDGP, 3, 4, "poisson", 0.0),
    (3, 2, 2, po.FE_DGQ, 3, 4, "adm", 0.0),
    (3, 1, 1, po.FE_DGQ, 3, 4, "poisson", 0.0),
    (2, 3, 2, po.FE_DGQ, 1, 3, "minsip", 0.0),
]


def variant(name, fe):
    return {
        "test": po.variant_minimal_sip_test,
        "adm": po.variant_assemble_dg_matrix,
        "poisson": lambda: po.variant_poisson_example(fe),
        "dr": lambda: po.variant_diffusion_reaction(fe),
        "minsip": po.variant_minimal_sip_example,
    }[name]()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dD_n%d_b%d_%s%d_q%d_%s_d%g" % (c[0], 2 ** c[1], c[2], c[3].name, c[4], c[5], c[6], c[7]))
def test_block_agglomeration_parity(case):
    dim, lg, b, fe_cls, p, nq, vname, dist = case
    fe = fe_cls(dim, p)
    ah = build(dim, lg, b, fe, nq, distort=dist)
    var = variant(vname, fe)
    rp, ci, ref = po.assemble_csr(ah, var, diag_first=True)
    got = gpu_values(flatten(ah, var, diag_first=True))
    assert_parity(got, ref, rp, ci, fe.n_dofs_per_cell)


TILED_CASES = [
    # (log2 cells/dir, block, fe ctor, degree, variant, distort, diag_first): more than 64 dofs per polytope (pdh_tiled.h)
    (1, 1, po.FE_DGQ, 4, "poisson", 0.1, True),      # n = 125: 2 x 2 tiles, the last one 61 functions wide
    (2, 2, po.FE_DGQ, 4, "dr", 0.1, False),          # 8-cell polytopes, reaction term, ascending (Trilinos) columns
    (2, 2, po.FE_DGQ, 4, "adm", 0.0, True),          # undistorted: must not be taken by a row kernel
    (1, 1, po.FE_AggloDGP, 6, "poisson", 0.15, True),  # n = 84
    (1, 1, po.FE_AggloDGP, 7, "dr", 0.0, False),     # n = 120, N1D = 8
    (1, 1, po.FE_DGQ, 5, "poisson", 0.1, True),      # n = 216: 4 x 4 tiles, the last one 24 wide
    (1, 1, po.FE_DGQ, 6, "adm", 0.1, True),          # n = 343: the largest degree examples/3D_piston.cc:912-914 runs
]


@pytest.mark.parametrize("case", TILED_CASES, ids=lambda c: "n%d_b%d_%s%d_%s_d%g_%s" % (2 ** c[0], c[1], c[2].name, c[3], c[4], c[5], "df" if c[6] else "asc"))
def test_more_than_64_dofs_per_polytope(case):
    """3-D FE_DGQ(4..6) / FE_AggloDGP(6, 7): blocks computed in 64 x 64 tiles (csrc/pdh_tiled.h), entry by entry against the oracle,
    both CSR layouts, with and without the reaction term, distorted and Cartesian cells; the right-hand side of the same problem too
    (reference examples/3D_piston.cc:912-914 sweeps degree 1 .. 6 with FE_DGQ<3>)."""
    import polydeal_amd as pa

    lg, b, fe_cls, p, vname, dist, diag_first = case
    fe = fe_cls(3, p)
    assert fe.n_dofs_per_cell > 64
    ah = build(3, lg, b, fe, p + 1, distort=dist)
    var = variant(vname, fe)
    rp, ci, ref = po.assemble_csr(ah, var, diag_first=diag_first)
    kw = flatten(ah, var, diag_first=diag_first, with_colind=True)
    ctx = pa.Context(0)
    ctx.set_problem(pa.Problem(**kw))
    assert ctx.algorithm_in_use() == "direct" and ctx.rows_kernel_in_use() == "none"
    got = ctx.assemble()
    assert_parity(got, ref, rp, ci, fe.n_dofs_per_cell)
    # right-hand side: volume source + Nitsche boundary datum (examples/poisson.cc:745-759, 788-828)
    f = lambda x: np.sin(2.0 * x[:, 0]) + x[:, 1] ** 2 + x[:, 2]
    g = lambda x: 1.0 + x[:, 0] * x[:, 1] - 0.5 * x[:, 2]
    ref_rhs = po.assemble_rhs(ah, var, f, g)
    got_rhs = ctx.assemble_rhs(f(kw["vq_x"].T), g(kw["fq_x"].T))
    ctx.close()
    assert np.max(np.abs(got_rhs - ref_rhs)) <= 1e-12 * np.max(np.abs(ref_rhs))


def test_irregular_agglomerates_and_ascending_layout():
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 3)
    ah = po.AgglomerationHandler(grid)
    gc.define_with_singletons(ah, grid.n_cells, gc.GROUPS_FOUR)
    fe = po.FE_AggloDGP(2, 2)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    for diag_first in (True, False):
        rp, ci, ref = po.assemble_csr(ah, var, diag_first=diag_first)
        got = gpu_values(flatten(ah, var, diag_first=diag_first, with_colind=diag_first))
        assert_parity(got, ref, rp, ci, fe.n_dofs_per_cell)


@pytest.mark.parametrize("dim", [2, 3])
def test_minimal_SIP_Poisson_identity_on_gpu(dim):
    """The reference's own known-answer test (test/polydeal/minimal_SIP_Poisson.cc:486-509) run through
    the HIP path: agglomerated blocks == one polytope per coarse cell, entry by entry to 1e-13."""
    fe = po.FE_DGQ(dim, 1)
    var = po.variant_minimal_sip_test()
    mats = []
    for agglomerated in (False, True):
        grid = po.hyper_cube_refined(dim, -1.0, 1.0, (2 if dim == 2 else 1) if agglomerated else (1 if dim == 2 else 0))
        ah = po.AgglomerationHandler(grid)
        if agglomerated:
            for g in (gc.GROUPS_2X2 if dim == 2 else [list(range(8))]):
                ah.define_agglomerate(g)
        else:
            for c in range(grid.n_cells):
                ah.define_agglomerate([c])
        ah.initialize_fe_values(3, 3)
        ah.distribute_agglomerated_dofs(fe)
        kw = flatten(ah, var)
        mats.append(po.csr_to_dense(kw["rowptr"], kw["colind"], gpu_values(kw), ah.n_dofs))
    assert np.max(np.abs(mats[0] - mats[1])) < 1e-13


def test_row_ranges_tile_the_matrix():
    """Multi-GPU formulation on one device: assembling two row ranges separately (as two ranks would)
    gives exactly the rows of the full assembly (owner-computes-rows, no exchange)."""
    import polydeal_amd as pa

    fe = po.FE_AggloDGP(3, 2)
    ah = build(3, 2, 2, fe, 3, distort=0.1)
    var = po.variant_diffusion_reaction(fe)
    kw = flatten(ah, var)
    full = gpu_values(kw)
    prob = pa.Problem(**kw)
    n = fe.n_dofs_per_cell
    nA = ah.n_agglomerates
    parts = []
    for r in range(3):
        ctx = pa.Context(0)
        ctx.set_problem(prob, (nA * r // 3) * n, (nA * (r + 1) // 3) * n)
        parts.append(ctx.assemble())
        ctx.close()
    # blocks of faces cut by the partition are computed from the other side there: equal to round-off
    got = np.concatenate(parts)
    assert got.shape == full.shape
    assert np.max(np.abs(got - full)) <= 1e-14 * np.max(np.abs(full))
    rp, ci, ref = po.assemble_csr(ah, var)
    assert_parity(full, ref, rp, ci, n)


def test_cpp_examples_run():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "minimal_SIP")
    # (the binaries are git-ignored build products: this is a GPU test, so a box that runs it without them has skipped build() -
    # that must show as a failure, not as one test fewer)
    assert os.path.exists(exe), "examples/ not built: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C examples)"
    # examples/minimal_SIP.cc on ITS mesh (meshes/t3.msh = tests/golden/t3.msh, refined 3 times, N = 50 .. 800 agglomerates): the
    # stdout is the reference's test/polydeal/poisson_sanity_check_03.output, up to the round-off sized "Test with 1" values
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "t3.msh")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got, want = out.stdout.split("\n"), gc.golden_lines("poisson_sanity_check_03.output")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w.startswith("Test with 1:"):
            assert g.startswith("Test with 1: ") and abs(float(g.split(":")[1])) < 1e-12
        else:
            assert g == w
    # examples/poisson.cc: p-convergence on t3.msh refined twice, 364 agglomerates, FE_AggloDGP(1..4), product_sine solution -
    # matrix, right-hand side, evaluation (interpolate_to_fine_grid) and error norms through the HIP path, CG on the host
    mesh = os.path.join(root, "tests", "golden", "t3.msh")
    out = subprocess.run([os.path.join(root, "examples", "poisson"), mesh], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.split("\n")
    assert lines[0] == "Testing p-convergence" and [l for l in lines if l.startswith("Fe degree: ")] == ["Fe degree: %d" % p for p in (1, 2, 3, 4)]
    assert sum(l.startswith("Time taken by assemble_system(): ") for l in lines) == 4
    l2 = [float(l.split(":")[1]) for l in lines if l.startswith("Error (L2):")]
    h1 = [float(l.split(":")[1]) for l in lines if l.startswith("Error (H1):")]
    nodal = [float(l.split()[-1]) for l in lines if l.startswith("interpolate_to_fine_grid:")]
    assert len(l2) == len(h1) == len(nodal) == 4
    # 364 polytopes of diameter ~0.1: sin(pi x) sin(pi y) is resolved better by a factor >= 4 (L2) / 2.5 (H1) per degree
    assert all(l2[k + 1] < 0.25 * l2[k] and h1[k + 1] < 0.4 * h1[k] for k in range(3)), (l2, h1)
    assert l2[0] < 5e-2 and l2[3] < 5e-7 and h1[3] < 5e-5 and nodal[3] < 5e-6, (l2, h1, nodal)
    out = subprocess.run([os.path.join(root, "examples", "poisson"), "--bench"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Assembled DoF/s" in out.stdout
    line = [l for l in out.stdout.split("\n") if l.startswith("compute_global_error")][0].replace("(", " ").replace(")", " ")
    t = line.split()
    assert abs(float(t[4]) - float(t[6])) < 1e-8 and abs(float(t[8]) - float(t[10])) < 1e-6, line
    # the reference's test minimal_SIP_Poisson.cc re-written against the C++ host mirror: its stdout must be the
    # reference's expected output file
    out = subprocess.run([os.path.join(root, "examples", "minimal_SIP_Poisson_test")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout == open(os.path.join(root, "tests", "golden", "minimal_SIP_Poisson.output")).read()


def test_cpp_diffusion_reaction_example_runs():
    """examples/diffusion_reaction.cc (BASELINE configs[3]'s caller) through the C ABI: FE_DGQ(1..4) - degree 4 with its 125 dofs per
    polytope through the tiled kernels, matrix, right-hand side and error sums -, reaction term, right-hand side
    and Nitsche datum of u = exp(xyz), the MPI ranks of the reference played in turn - every rank with its rank-local description
    in Epetra column order assembling only its rows -, host CG, error norms summed over the ranks: p-convergence."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "diffusion_reaction")
    assert os.path.exists(exe), "examples/ not built: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C examples)"
    for ranks in ("4", "1"):
        out = subprocess.run([exe, "3", ranks, "10" if ranks == "4" else "40"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.split("\n")
        l2 = [float(l.split(":")[1]) for l in lines if l.startswith("L2 error (exponential solution):")]
        h1 = [float(l.split(":")[1]) for l in lines if l.startswith("Semi H1 error (exponential solution):")]
        assert len(l2) == len(h1) == 4 and sum(l.startswith("Time taken by assemble_system()") for l in lines) == 4
        assert all(l2[k + 1] < l2[k] / 3.0 and h1[k + 1] < h1[k] / 3.0 for k in range(3)), (l2, h1)
        assert l2[0] < 2e-2 and l2[2] < 1e-4 and h1[2] < 5e-3 and l2[3] < 1e-5, (l2, h1)
        if ranks == "4":
            first = (l2, h1)
    # the same 40 agglomerates (same seed) assembled by one rank: the split into ranks changes nothing but rounding
    assert all(abs(a - b) <= 1e-9 * a for a, b in zip(first[0] + first[1], l2 + h1)), (first, l2, h1)


def test_poisson_output_L2_error_with_gpu_matrix():
    """test/polydeal/poisson.output ('L2 error:0.00647702', printed by the reference itself): same pipeline
    as tests/test_oracle_golden.py::test_poisson_output_L2_error but with the matrix assembled by the HIP path."""
    from test_oracle_golden import _poisson_test_setup, poisson_l2_error

    grid, ah, var = _poisson_test_setup()
    kw = flatten(ah, var, diag_first=False)
    vals = gpu_values(kw)
    rp, ci, ref = po.assemble_csr(ah, var, diag_first=False)
    assert_parity(vals, ref, rp, ci, ah.fe.n_dofs_per_cell)
    err = poisson_l2_error(grid, ah, kw["rowptr"], kw["colind"], vals)
    assert "L2 error:" + gc.fmt(err) == gc.golden_lines("poisson.output")[0]


def test_multilevel_block_hierarchy():
    """configs[4] stand-in (meshes/piston_3.inp is not in the reference snapshot): R-tree-like block levels
    of a jittered 16^3 grid, one assemble_dg_matrix per level, each checked against the oracle."""
    import polydeal_amd as pa
    from polydeal_amd.levels import assemble_levels, block_hierarchy

    grid = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 3).distort(0.2, seed=7)
    fe = pa.FE_DGQ(3, 1)
    levels = block_hierarchy(grid, fe, [4, 2, 1])
    mats = assemble_levels(levels, fe)
    assert [ah.n_agglomerates for ah in levels] == [8, 64, 512]
    og = po.hyper_cube_refined(3, 0.0, 1.0, 3)
    for c in range(og.n_cells):
        og.vertices[c] = grid.cell_vertices(c)
    for b, (rp, ci, vals) in zip([4, 2, 1], mats):
        oah = po.AgglomerationHandler(og)
        for g in po.block_agglomerates(og, b):
            oah.define_agglomerate(g)
        oah.initialize_fe_values(2, 2)
        oah.distribute_agglomerated_dofs(po.FE_DGQ(3, 1))
        orp, oci, ref = po.assemble_csr(oah, po.variant_assemble_dg_matrix())
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        assert_parity(vals, ref, orp, oci, fe.n_dofs_per_cell)


@pytest.mark.parametrize("basis,p,blocks", [("dgq", 1, [4, 2, 1]), ("dgp", 3, [4, 2]), ("dgq", 3, [2, 1])])
def test_multilevel_block_hierarchy_on_a_cartesian_grid(basis, p, blocks):
    """The same per-level assembly on an UNDISTORTED grid: assemble_levels hands every level over without its quadrature points
    (generated on the device) where the term kernels take it, on one context for all levels; the level of 4^3-cell polytopes at degree
    3 exceeds their LDS budget and silently takes the points-based description.  Every level against the oracle."""
    import polydeal_amd as pa
    from polydeal_amd.levels import assemble_levels, block_hierarchy

    grid = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 3)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(3, p)
    levels = block_hierarchy(grid, fe, blocks)
    mats = assemble_levels(levels, fe, pa.SipVariant.poisson_example(fe))
    og = po.hyper_cube_refined(3, 0.0, 1.0, 3)
    ofe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    for b, (rp, ci, vals) in zip(blocks, mats):
        oah = po.AgglomerationHandler(og)
        for g in po.block_agglomerates(og, b):
            oah.define_agglomerate(g)
        oah.initialize_fe_values(p + 1, p + 1)
        oah.distribute_agglomerated_dofs(ofe)
        orp, oci, ref = po.assemble_csr(oah, po.variant_poisson_example(ofe))
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        assert_parity(vals, ref, orp, oci, fe.n_dofs_per_cell)


@pytest.mark.parametrize("dim,lg,b,fe_cls,p,dist", [
    (2, 3, 2, po.FE_DGQ, 1, 0.0),
    (2, 3, 2, po.FE_AggloDGP, 3, 0.2),
    (3, 2, 2, po.FE_DGQ, 3, 0.0),
    (3, 2, 2, po.FE_AggloDGP, 2, 0.15),
])
def test_rhs_parity(dim, lg, b, fe_cls, p, dist):
    """pdh_assemble_rhs vs the oracle: volume source + Nitsche boundary datum (examples/poisson.cc:745-759,
    788-828), 1e-12 relative."""
    import polydeal_amd as pa

    fe = fe_cls(dim, p)
    ah = build(dim, lg, b, fe, p + 1, distort=dist)
    var = po.variant_poisson_example(fe)
    f = lambda x: np.sin(2.0 * x[:, 0]) + x[:, 1] ** 2 + (x[:, -1] if dim == 3 else 0.0)
    g = lambda x: 1.0 + x[:, 0] * x[:, 1] - 0.5 * x[:, -1]
    kw = flatten(ah, var)
    nq = kw["vq_x"].shape[1]
    f_vol = f(kw["vq_x"].T)
    g_b = g(kw["fq_x"].T)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    got = ctx.assemble_rhs(f_vol, g_b)
    got_f = ctx.assemble_rhs(f_vol, None)
    ctx.close()
    ref = po.assemble_rhs(ah, var, f, g)
    ref_f = po.assemble_rhs(ah, var, f, None)
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))
    assert np.max(np.abs(got_f - ref_f)) <= TOL * np.max(np.abs(ref_f))


def test_poisson_output_L2_error_all_on_gpu():
    """The reference's printed 'L2 error:0.00647702' (test/polydeal/poisson.output) with BOTH the matrix and
    the right-hand side assembled by the HIP path (solve + error evaluation on the host)."""
    import polydeal_amd as pa
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from test_oracle_golden import _poisson_test_setup

    grid, ah, var = _poisson_test_setup()
    kw = flatten(ah, var, diag_first=False)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    vals = ctx.assemble()
    pi = np.pi
    xq = kw["vq_x"]
    b = ctx.assemble_rhs(8 * pi * pi * np.sin(2 * pi * xq[0]) * np.sin(2 * pi * xq[1]), None)
    ctx.close()
    A = sp.csr_matrix((vals, kw["colind"], kw["rowptr"]), shape=(ah.n_dofs, ah.n_dofs))
    u = spla.spsolve(A.tocsc(), b)
    err2 = 0.0
    for P in range(ah.n_agglomerates):
        coef = u[ah.dof_indices(P)]
        for cell in ah.get_agglomerate(P):
            V = grid.vertices[cell]
            val, _ = ah.fe.shape(ah.real_to_unit(P, V))
            mid = V.mean(axis=0)
            h2 = (V[1, 0] - V[0, 0]) * (V[2, 1] - V[0, 1])
            err2 += h2 * (np.mean(val @ coef) - np.sin(2 * pi * mid[0]) * np.sin(2 * pi * mid[1])) ** 2
    assert "L2 error:" + gc.fmt(np.sqrt(err2)) == gc.golden_lines("poisson.output")[0]


ALL_COMBOS = [(2, po.FE_DGQ, p) for p in range(0, 8)] + [(2, po.FE_AggloDGP, p) for p in range(1, 8)] + \
             [(3, po.FE_DGQ, p) for p in range(0, 4)] + [(3, po.FE_AggloDGP, p) for p in range(1, 6)]


@pytest.mark.parametrize("dim,fe_cls,p", ALL_COMBOS, ids=lambda v: getattr(v, "name", str(v)))
def test_every_instantiated_combo(dim, fe_cls, p):
    """Every (dim, basis, degree) the library instantiates kernels for (pdh_combos.h), on a small distorted
    mesh with the reaction term switched on for odd p (exercises both k_diag variants)."""
    fe = fe_cls(dim, p)
    ah = build(dim, 2, 2, fe, p + 1, distort=0.1)
    var = po.variant_diffusion_reaction(fe) if p % 2 else po.variant_assemble_dg_matrix()
    if p == 0:
        var = po.SipVariant("p0", 10.0, "id", "diameter_in")
    rp, ci, ref = po.assemble_csr(ah, var)
    got = gpu_values(flatten(ah, var))
    assert_parity(got, ref, rp, ci, fe.n_dofs_per_cell)


def random_agglomeration(grid, n_seeds, rng, allow_disconnected=False):
    """Random polytopes: grow n_seeds regions over the cell adjacency graph (round-robin BFS with random
    frontier picks); optionally glue two random regions into one disconnected agglomerate
    (test/polydeal/disconnected_exact_solution.cc exercises such agglomerates)."""
    n = grid.n_cells
    owner = -np.ones(n, dtype=np.int64)
    seeds = rng.choice(n, size=n_seeds, replace=False)
    frontiers = []
    for k, s in enumerate(seeds):
        owner[s] = k
        frontiers.append([int(s)])
    remaining = n - n_seeds
    while remaining:
        progressed = False
        for k in rng.permutation(n_seeds):
            fr = frontiers[k]
            while fr:
                c = fr[rng.integers(len(fr))]
                nbrs = [grid.neighbor(c, f) for f in range(2 * grid.dim)]
                free = [x for x in nbrs if x != po.INVALID and owner[x] < 0]
                if not free:
                    fr.remove(c)
                    continue
                x = free[rng.integers(len(free))]
                owner[x] = k
                fr.append(int(x))
                remaining -= 1
                progressed = True
                break
        assert progressed or remaining == 0
    groups = [sorted(np.nonzero(owner == k)[0].tolist()) for k in range(n_seeds)]
    if allow_disconnected and n_seeds > 3:
        a, b = 0, n_seeds - 1
        groups[a] = sorted(groups[a] + groups[b])
        del groups[b]
    return groups


@pytest.mark.parametrize("dim,lg,n_seeds,fe_cls,p,seed,disc", [
    (2, 4, 23, po.FE_AggloDGP, 2, 11, False),
    (2, 4, 9, po.FE_DGQ, 1, 12, True),
    (2, 3, 7, po.FE_DGQ, 3, 13, False),
    (3, 2, 5, po.FE_AggloDGP, 3, 14, False),
    (3, 2, 9, po.FE_DGQ, 2, 15, True),
    (3, 3, 40, po.FE_AggloDGP, 1, 16, False),
])
def test_random_irregular_agglomerates(dim, lg, n_seeds, fe_cls, p, seed, disc):
    """Irregular polytopes (varying sub-cell counts, many faces per polytope, non-convex and disconnected
    shapes, un-sorted definition order) on a distorted mesh - the general input the face tables must handle."""
    rng = np.random.default_rng(seed)
    grid = po.hyper_cube_refined(dim, -1.0, 1.0, lg).distort(0.2, seed=seed)
    groups = random_agglomeration(grid, n_seeds, rng, disc)
    order = rng.permutation(len(groups))  # polytope index order != master cell order
    ah = po.AgglomerationHandler(grid)
    for k in order:
        ah.define_agglomerate(groups[k])
    fe = fe_cls(dim, p)
    ah.initialize_fe_values(p + 1, p + 1)
    ah.distribute_agglomerated_dofs(fe)
    assert max(ah.n_faces) > 2 * dim or n_seeds < 8
    for var in (po.variant_poisson_example(fe), po.variant_assemble_dg_matrix()):
        rp, ci, ref = po.assemble_csr(ah, var)
        got = gpu_values(flatten(ah, var))
        assert_parity(got, ref, rp, ci, fe.n_dofs_per_cell)


@pytest.mark.parametrize("dim,lg,b,fe_cls,p,dist", [
    (2, 3, 2, po.FE_DGQ, 2, 0.2),
    (2, 3, 4, po.FE_AggloDGP, 4, 0.0),
    (3, 2, 2, po.FE_DGQ, 3, 0.1),
    (3, 2, 2, po.FE_AggloDGP, 3, 0.0),
    (3, 1, 1, po.FE_DGQ, 4, 0.1),       # n = 125 > 64
    (3, 1, 1, po.FE_AggloDGP, 7, 0.0),  # n = 120, N1D = 8
])
def test_evaluate_and_global_error_parity(dim, lg, b, fe_cls, p, dist):
    """pdh_evaluate vs the oracle (values and gradients at the volume quadrature points, 1e-12 relative) and
    PolyUtils::compute_global_error (include/poly_utils.h:1647-1750) rebuilt on top of it; also on a row range."""
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    fe = fe_cls(dim, p)
    ah = build(dim, lg, b, fe, p + 1, distort=dist)
    var = po.variant_poisson_example(fe)
    kw = flatten(ah, var)
    rng = np.random.default_rng(5)
    u = rng.standard_normal(ah.n_dofs)
    exact = lambda x: np.sin(1.3 * x[:, 0]) * np.cos(0.7 * x[:, 1]) + (x[:, -1] ** 2 if dim == 3 else 0.0)

    def exact_grad(x):
        g = np.zeros_like(x)
        g[:, 0] = 1.3 * np.cos(1.3 * x[:, 0]) * np.cos(0.7 * x[:, 1])
        g[:, 1] = -0.7 * np.sin(1.3 * x[:, 0]) * np.sin(0.7 * x[:, 1])
        if dim == 3:
            g[:, 2] = 2 * x[:, 2]
        return g

    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    uh, gh = ctx.evaluate(u, kw["vq_ptr"], kw["vq_x"], want_grad=True)
    uh_only = ctx.evaluate(u, kw["vq_ptr"], kw["vq_x"])
    l2, h1 = pa.compute_global_error(ctx, kw["vq_ptr"], kw["vq_x"], kw["vq_w"], u, exact, exact_grad)
    ref_u = np.concatenate([po.evaluate_at(ah, u, P, ah.reinit(P)["x"])[0] for P in range(ah.n_agglomerates)])
    ref_g = np.concatenate([po.evaluate_at(ah, u, P, ah.reinit(P)["x"])[1] for P in range(ah.n_agglomerates)]).T
    assert np.array_equal(uh, uh_only)
    assert np.max(np.abs(uh - ref_u)) <= TOL * np.max(np.abs(ref_u))
    assert np.max(np.abs(gh - ref_g)) <= TOL * np.max(np.abs(ref_g))
    rl2, rh1 = po.compute_global_error(ah, u, exact, exact_grad)
    assert abs(l2 - rl2) <= TOL * rl2 and abs(h1 - rh1) <= TOL * rh1
    # the sums are formed on the device (pdh_global_error: 16 bytes per polytope come back): same numbers as summing the
    # evaluated values here, and reproducible bit for bit
    xs = np.asarray(kw["vq_x"]).T
    eu, eg = exact(xs), exact_grad(xs).T
    s_dev = ctx.global_error_sums(u, kw["vq_ptr"], kw["vq_x"], kw["vq_w"], eu, eg)
    s_host = (float(np.sum(kw["vq_w"] * (eu - uh) ** 2)), float(np.sum(kw["vq_w"] * np.sum((eg - gh) ** 2, axis=0))))
    assert abs(s_dev[0] - s_host[0]) <= 1e-13 * s_host[0] and abs(s_dev[1] - s_host[1]) <= 1e-13 * s_host[1]
    assert s_dev == ctx.global_error_sums(u, kw["vq_ptr"], kw["vq_x"], kw["vq_w"], eu, eg)

    # two "ranks": squares of the partial errors add up (poly_utils.h:1736-1745)
    s2 = np.zeros(2)
    for r in range(2):
        rb, re = row_range(ah.n_agglomerates, fe.n_dofs_per_cell, r, 2)
        ctx.set_problem(prob, rb, re)
        a, c = pa.compute_global_error(ctx, kw["vq_ptr"], kw["vq_x"], kw["vq_w"], u[rb:re], exact, exact_grad)
        s2 += [a * a, c * c]
    ctx.close()
    assert abs(np.sqrt(s2[0]) - rl2) <= TOL * rl2 and abs(np.sqrt(s2[1]) - rh1) <= TOL * rh1


def test_poisson_output_L2_error_evaluated_on_gpu():
    """'L2 error:0.00647702' (test/polydeal/poisson.output) with matrix, right-hand side AND the
    interpolate_to_fine_grid evaluation (u_h at the vertices of every sub-cell, include/poly_utils.h:1196-1233)
    on the HIP path; only the sparse solve and the one-point integrate_difference sum stay on the host."""
    import polydeal_amd as pa
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from test_oracle_golden import _poisson_test_setup

    grid, ah, var = _poisson_test_setup()
    kw = flatten(ah, var, diag_first=False)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    vals = ctx.assemble()
    pi = np.pi
    xq = kw["vq_x"]
    b = ctx.assemble_rhs(8 * pi * pi * np.sin(2 * pi * xq[0]) * np.sin(2 * pi * xq[1]), None)
    A = sp.csr_matrix((vals, kw["colind"], kw["rowptr"]), shape=(ah.n_dofs, ah.n_dofs))
    u = spla.spsolve(A.tocsc(), b)
    cells = [ah.get_agglomerate(P) for P in range(ah.n_agglomerates)]
    pt_ptr = np.concatenate([[0], np.cumsum([4 * len(c) for c in cells])])
    pts = np.concatenate([grid.vertices[c] for cs in cells for c in cs]).T  # [2][4*n_cells]
    uv = pa.interpolate_to_points(ctx, u, pt_ptr, pts).reshape(-1, 4)
    ctx.close()
    V = np.stack([grid.vertices[c] for cs in cells for c in cs])  # [n_cells,4,2]
    mid = V.mean(axis=1)
    h2 = (V[:, 1, 0] - V[:, 0, 0]) * (V[:, 2, 1] - V[:, 0, 1])
    err = np.sqrt(np.sum(h2 * (uv.mean(axis=1) - np.sin(2 * pi * mid[:, 0]) * np.sin(2 * pi * mid[:, 1])) ** 2))
    assert "L2 error:" + gc.fmt(err) == gc.golden_lines("poisson.output")[0]


def test_multilevel_block_hierarchy_p3():
    """configs[4] stand-in at ITS degree (meshes/piston_3.inp is not in the reference snapshot): three R-tree-like block
    levels (64, 8 and 1 cells per polytope) of a jittered 8^3 grid with FE_DGQ(3), one assemble_dg_matrix per level
    (examples/simplex_agglomerated_multigrid.cc:378-390), every level against the oracle entry by entry."""
    import polydeal_amd as pa
    from polydeal_amd.levels import assemble_levels, block_hierarchy

    grid = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 3).distort(0.15, seed=11)
    fe = pa.FE_DGQ(3, 3)
    levels = block_hierarchy(grid, fe, [4, 2, 1])
    mats = assemble_levels(levels, fe)
    assert [ah.n_agglomerates for ah in levels] == [8, 64, 512]
    og = po.hyper_cube_refined(3, 0.0, 1.0, 3)
    for c in range(og.n_cells):
        og.vertices[c] = grid.cell_vertices(c)
    for b, (rp, ci, vals) in zip([4, 2, 1], mats):
        oah = po.AgglomerationHandler(og)
        for g in po.block_agglomerates(og, b):
            oah.define_agglomerate(g)
        oah.initialize_fe_values(4, 4)
        oah.distribute_agglomerated_dofs(po.FE_DGQ(3, 3))
        orp, oci, ref = po.assemble_csr(oah, po.variant_assemble_dg_matrix())
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        assert_parity(vals, ref, orp, oci, fe.n_dofs_per_cell)
    # the same hierarchy on the undistorted grid runs through the row kernel: levels must agree with the moment form
    grid2 = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 3)
    for ah in block_hierarchy(grid2, fe, [4, 2, 1]):
        flat = ah.flatten(pa.SipVariant.assemble_dg_matrix(), True, False)
        out = {}
        for alg in ("rows", "moment"):
            ctx = pa.Context(0)
            ctx.set_algorithm(alg)
            ctx.set_problem(flat)
            assert ctx.algorithm_in_use() == alg
            out[alg] = ctx.assemble()
            ctx.close()
        assert np.max(np.abs(out["rows"] - out["moment"])) <= 1e-13 * np.max(np.abs(out["moment"]))


@pytest.mark.parametrize("dim,lg,bc,bf,p,dist", [
    (2, 3, 4, 2, 1, 0.0), (2, 3, 4, 1, 3, 0.2), (2, 4, 8, 2, 7, 0.1), (3, 2, 4, 2, 2, 0.1), (3, 2, 2, 1, 3, 0.0),
    (3, 3, 4, 2, 3, 0.15), (3, 3, 2, 1, 3, 0.15),  # the level pairs of the p = 3 hierarchy below (configs[4] stand-in)
    (3, 2, 2, 1, 4, 0.1),  # n = 125 > 64
])
def test_injection_matrix_parity(dim, lg, bc, bf, p, dist):
    """Utils::fill_injection_matrix (include/utils.h:95-270) through the host mirror + pdh_shape_values vs the
    oracle, 1e-12 relative, and the property of test/polydeal/distributed_injection_01.cc."""
    import polydeal_amd as pa

    fe_o = po.FE_DGQ(dim, p)
    grid_o = po.hyper_cube_refined(dim, 0.0, 1.0, lg)
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, lg)
    if dist:  # the two RNG streams differ: distort the product's grid and copy its vertices
        grid.distort(dist, seed=3)
        for cell in range(grid_o.n_cells):
            grid_o.vertices[cell] = grid.cell_vertices(cell)
    handlers, oracles = [], []
    for b in (bc, bf):
        ah = pa.AgglomerationHandler(grid)
        ah.define_block_agglomerates(b)
        ah.initialize_fe_values(p + 1, p + 1)
        ah.distribute_agglomerated_dofs(pa.FE_DGQ(dim, p))
        handlers.append(ah)
        ao = po.AgglomerationHandler(grid_o)
        for g in po.block_agglomerates(grid_o, b):
            ao.define_agglomerate(g)
        ao.initialize_fe_values(p + 1, p + 1)
        ao.distribute_agglomerated_dofs(fe_o)
        oracles.append(ao)
    rp, ci, va = pa.fill_injection_matrix(handlers[0], handlers[1])
    n = fe_o.n_dofs_per_cell
    assert np.array_equal(rp, np.arange(len(rp)) * n)
    M = po.fill_injection_matrix(oracles[0], oracles[1])
    got = np.zeros_like(M)
    rows = np.repeat(np.arange(len(rp) - 1), n)
    got[rows, ci] = va
    assert np.max(np.abs(got - M)) <= TOL * np.max(np.abs(M))
    f = lambda x: 1.0 + x[:, 0] ** p - 0.5 * x[:, 1] ** p * x[:, 0] + (x[:, -1] ** p if dim == 3 else 0.0)
    err = got @ po.interpolate_nodal(oracles[0], f) - po.interpolate_nodal(oracles[1], f)
    assert np.max(np.abs(err)) < 5e-13
    with pytest.raises(pa.HostError):
        pa.fill_injection_matrix(handlers[1], handlers[0])  # coarse must be smaller (utils.h:120)


def _dense_from_gpu(ah, var, diag_first=True):
    kw = flatten(ah, var, diag_first=diag_first)
    vals = gpu_values(kw)
    return po.csr_to_dense(kw["rowptr"], kw["colind"], vals, ah.n_dofs), kw


@pytest.mark.parametrize("dim,fe_cls,p", [(2, po.FE_DGQ, 2), (3, po.FE_AggloDGP, 2), (3, po.FE_DGQ, 3)])
def test_single_polytope_has_no_interior_faces(dim, fe_cls, p):
    """Edge case: the whole mesh is one polytope - zero coupling items, one boundary face made of every boundary
    sub-face (agglomeration_handler.cc:1575-1577).  Also the other extreme: every cell its own polytope."""
    fe = fe_cls(dim, p)
    lg = 2 if dim == 2 else 1
    ah = build(dim, lg, 2 ** lg, fe, p + 1)
    assert ah.n_agglomerates == 1 and ah.n_faces_of(0) == 1
    var = po.variant_poisson_example(fe)
    A, kw = _dense_from_gpu(ah, var)
    assert kw["n_faces"] == 1
    ref = po.assemble_dense(ah, var)
    assert np.max(np.abs(A - ref)) <= TOL * np.max(np.abs(ref))
    ah1 = build(dim, lg, 1, fe, p + 1, distort=0.1)
    A1, _ = _dense_from_gpu(ah1, var, diag_first=False)
    ref1 = po.assemble_dense(ah1, var)
    assert np.max(np.abs(A1 - ref1)) <= TOL * np.max(np.abs(ref1))


@pytest.mark.parametrize("groups", [gc.GROUPS_FOUR, gc.GROUPS_QUAD_PTS])
def test_poisson_sanity_identities_on_gpu(groups):
    """test/polydeal/poisson_sanity_check_03.output (v^T A v = 1, 2, ~1e-14 for v = x, x+y, 1 on ANY agglomeration
    of the unit square, boundary terms dropped) with the matrix from the HIP path."""
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 3)
    ah = po.AgglomerationHandler(grid)
    gc.define_with_singletons(ah, grid.n_cells, groups)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    var = po.SipVariant("sanity", 10.0, "index", "diameter_in", boundary="zero")
    A, _ = _dense_from_gpu(ah, var)
    forms = [float(v @ A @ v) for v in (po.interpolate_nodal(ah, f) for f in
                                         (lambda x: x[:, 0], lambda x: x[:, 0] + x[:, 1], lambda x: np.ones(len(x))))]
    assert abs(forms[0] - 1.0) < 1e-12 and abs(forms[1] - 2.0) < 1e-12 and abs(forms[2]) < 1e-12
    lines = gc.golden_lines("poisson_sanity_check_03.output")
    assert "Test with f(x,y)=x:" + gc.fmt(round(forms[0], 12)) == lines[1]
    assert "Test with f(x,y)=x+y:" + gc.fmt(round(forms[1], 12)) == lines[2]


@pytest.mark.parametrize("fe_cls,p", [(po.FE_DGQ, 1), (po.FE_AggloDGP, 1), (po.FE_AggloDGP, 2), (po.FE_DGQ, 2)])
def test_exact_solution_reproduced_all_on_gpu(fe_cls, p):
    """test/polydeal/exact_solutions_dgp.cc:685-704: a solution inside the polytopal space is reproduced to
    ~1e-14 on a randomly distorted grid - matrix, Nitsche right-hand side and the error evaluation on the HIP
    path (dense solve on the host)."""
    import polydeal_amd as pa

    grid = po.hyper_cube_refined(2, 0.0, 1.0, 2).distort(0.25, seed=1)
    ah = po.AgglomerationHandler(grid)
    for g in [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]:
        ah.define_agglomerate(g)
    fe = fe_cls(2, p)
    ah.initialize_fe_values(2 * p + 1, 2 * p + 1)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    u_ex = (lambda x: x[:, 0] + x[:, 1] - 1.0) if p == 1 else (lambda x: x[:, 0] ** 2 + x[:, 0] * x[:, 1])
    lap = 0.0 if p == 1 else -2.0
    kw = flatten(ah, var)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    A = po.csr_to_dense(kw["rowptr"], kw["colind"], ctx.assemble(), ah.n_dofs)
    b = ctx.assemble_rhs(np.full(kw["vq_x"].shape[1], lap), u_ex(kw["fq_x"].T))
    u = np.linalg.solve(A, b)
    l2, _ = pa.compute_global_error(ctx, kw["vq_ptr"], kw["vq_x"], kw["vq_w"], u, u_ex)
    ctx.close()
    assert l2 < 1e-12
    assert np.max(np.abs(A - A.T)) < 1e-11 * np.max(np.abs(A))


def _values(kw, alg, r0=0, r1=None):
    import polydeal_amd as pa

    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_algorithm(alg)
    ctx.set_problem(prob, r0, r1)
    used = ctx.algorithm_in_use()
    v = ctx.assemble()
    ctx.close()
    return v, used


def _values_k(kw, alg, r0=0, r1=None, terms=True):
    """values, algorithm in use, row kernel in use; terms=False keeps the kinds of pdh_rows.h (PDH_TERMS=0, read per set_problem)"""
    import os

    import polydeal_amd as pa

    old = os.environ.get("PDH_TERMS")
    os.environ["PDH_TERMS"] = "1" if terms else "0"
    try:
        prob = pa.Problem(**kw)
        ctx = pa.Context(0)
        ctx.set_algorithm(alg)
        ctx.set_problem(prob, r0, r1)
        used, kern = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
        v = ctx.assemble()
        ctx.close()
    finally:
        if old is None:
            del os.environ["PDH_TERMS"]
        else:
            os.environ["PDH_TERMS"] = old
    return v, used, kern


MOMENT_CASES = [
    # fe, degree, log2 cells/dir, block, distort, variant, diag_first
    (po.FE_DGQ, 3, 2, 2, 0.0, "poisson", True),
    (po.FE_DGQ, 3, 2, 2, 0.2, "dr", True),
    (po.FE_DGQ, 3, 2, 2, 0.1, "adm", False),
    (po.FE_DGQ, 3, 1, 2, 0.0, "poisson", True),   # one polytope: no coupling items
    (po.FE_DGQ, 3, 1, 1, 0.1, "test", True),      # every cell its own polytope
    (po.FE_DGQ, 2, 2, 2, 0.15, "dr", True),
    (po.FE_DGQ, 1, 2, 2, 0.1, "minsip", True),
    (po.FE_AggloDGP, 3, 2, 2, 0.1, "poisson", True),
    (po.FE_AggloDGP, 2, 2, 1, 0.0, "adm", False),
    (po.FE_AggloDGP, 1, 2, 2, 0.2, "poisson", True),
]


@pytest.mark.parametrize("fe_cls,p,lg,b,dist,varname,diag_first", MOMENT_CASES,
                         ids=lambda v: getattr(v, "name", str(v)))
def test_moment_form_parity(fe_cls, p, lg, b, dist, varname, diag_first):
    """The moment form (csrc/pdh_moment.h: Legendre moments of the quadrature + sum factorisation) against the oracle,
    1e-12 relative, next to the direct form on the same inputs - the two differ by rounding only."""
    fe = fe_cls(3, p)
    ah = build(3, lg, b, fe, p + 1, distort=dist)
    var = variant(varname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    vm, used = _values(kw, "moment")
    vd, used_d = _values(kw, "direct")
    assert used == "moment" and used_d == "direct"
    sc = np.max(np.abs(ref))
    assert_parity_ah(vm, ref, ah, diag_first, what="moment")
    assert_parity_ah(vd, ref, ah, diag_first, what="direct")
    assert np.max(np.abs(vm - vd)) <= 1e-13 * sc
    # AUTO: undistorted (every face an axis-aligned plane, tensor rules) -> row kernel; else the moment
    # form for both kinds of block at FE_DGQ(3), for the diagonal blocks only at FE_DGQ(2), else direct
    va, used_a = _values(kw, "auto")
    expect = "direct"
    if dist == 0.0:
        expect = "rows"
    elif fe_cls is po.FE_DGQ and p == 3:
        expect = "moment"
    elif fe_cls is po.FE_DGQ and p == 2:
        expect = "mixed"
    assert used_a == expect
    assert_parity_ah(va, ref, ah, diag_first, what="auto")


@pytest.mark.parametrize("seed,disc", [(0, False), (1, True)])
def test_moment_form_irregular_agglomerates_and_row_ranges(seed, disc):
    """Moment form on random irregular (also disconnected) agglomerates with ragged point counts, and on row ranges:
    a face cut by the partition is contracted from the side that is owned (coupling item with P = side 1)."""
    from polydeal_amd.partition import row_range

    rng = np.random.default_rng(seed)
    fe = po.FE_DGQ(3, 3)
    grid = po.hyper_cube_refined(3, 0.0, 1.0, 2)
    grid.distort(0.15, seed=seed)
    ah = po.AgglomerationHandler(grid)
    for g in random_agglomeration(grid, 9, rng, allow_disconnected=disc):
        ah.define_agglomerate(g)
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    kw = flatten(ah, var)
    ref = po.assemble_csr(ah, var)[2]
    sc = np.max(np.abs(ref))
    vm, _ = _values(kw, "moment")
    assert_parity_ah(vm, ref, ah)
    n = fe.n_dofs_per_cell
    parts = []
    for r in range(3):
        rb, re = row_range(ah.n_agglomerates, n, r, 3)
        parts.append(_values(kw, "moment", rb, re)[0])
    assert_parity_ah(np.concatenate(parts), ref, ah)


def test_moment_form_unavailable_is_reported():
    """2-D problems have no moment form: forcing it must fail loudly, AUTO must fall back to the direct kernels."""
    import polydeal_amd as pa

    fe = po.FE_DGQ(2, 2)
    ah = build(2, 3, 2, fe, 3)
    kw = flatten(ah, po.variant_poisson_example(fe))
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    assert ctx.algorithm_in_use() == "direct"
    ctx.set_algorithm("moment")
    with pytest.raises(pa.PdhError):
        ctx.assemble()
    ctx.close()


def test_overlapped_and_serial_launch_agree():
    """Large problems run their two kernels concurrently on two streams (pdh_set_overlap); the values - every CSR entry is
    written by exactly one work item - must be bitwise those of the serial launch, with either algorithm."""
    import polydeal_amd as pa

    fe = po.FE_DGQ(3, 1)
    ah = build(3, 4, 1, fe, 2)  # 4096 polytopes + 11520 interior faces: above the overlap threshold
    kw = flatten(ah, po.variant_poisson_example(fe), with_colind=False)
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    out = {}
    for alg in ("direct", "moment"):
        ctx.set_algorithm(alg)
        for ov in (True, False):
            ctx.set_overlap(ov)
            out[(alg, ov)] = ctx.assemble()
        assert np.array_equal(out[(alg, True)], out[(alg, False)])
    ctx.set_overlap(True)
    ctx.set_profiling(True)
    ctx.assemble_device()
    (t0, t1), nl = ctx.kernel_times_ms()
    ctx.close()
    assert nl == 1 and t0 > 0 and t1 > 0
    sc = np.max(np.abs(out[("direct", False)]))
    assert np.max(np.abs(out[("direct", False)] - out[("moment", False)])) <= 1e-13 * sc


@pytest.mark.parametrize("r,lg", [(2, 2), (4, 3), (8, 4)])
def test_moment_form_with_neighbours_of_very_different_size(r, lg):
    """One r^3-cell polytope (index 0, so that its faces are processed from ITS side) among single-cell polytopes.
    The face moments are taken per direction in the shorter of the two bounding-box intervals; in the big polytope's own
    frame the mixed 1-D tables cancel catastrophically (measured 1e-9 relative at r = 4, 3e-7 at r = 8)."""
    fe = po.FE_DGQ(3, 3)
    grid = po.hyper_cube_refined(3, 0.0, 1.0, lg)
    ah = po.AgglomerationHandler(grid)
    big = sorted(int(grid.ijk_to_cell[(i, j, k)]) for i in range(r) for j in range(r) for k in range(r))
    ah.define_agglomerate(big)
    for c in range(grid.n_cells):
        if c not in set(big):
            ah.define_agglomerate([c])
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    kw = flatten(ah, po.variant_poisson_example(fe), with_colind=False)
    vd, _ = _values(kw, "direct")
    vm, _ = _values(kw, "moment")
    assert np.max(np.abs(vm - vd)) <= 1e-12 * np.max(np.abs(vd))
    # AUTO: the row kernel where the big polytope's face entries fit (MULTI instantiation: every small neighbour is an entry of
    # its own; with tensor rules both bases are evaluated in their own boxes - no common frame at all), else the forms above
    va, used = _values(kw, "auto")
    assert used in ("rows", "moment")
    assert np.max(np.abs(va - vd)) <= 1e-12 * np.max(np.abs(vd))
    if used == "rows":  # the same with the general-point paths (2-D moments in the shorter frame)
        vg, used_g = _values(dict(kw, vq_tensor_n=-1, fq_tensor_n=-1), "rows")
        assert used_g == "rows"
        assert np.max(np.abs(vg - vd)) <= 1e-12 * np.max(np.abs(vd))


# ---------------------------------------------------------------------------------------------------------------------
# The PRODUCT's own chain (C++ host mirror -> pdh_problem -> HIP kernels) against the oracle, entry by entry: the cases
# above feed the kernels from the oracle's flattening (identical inputs); here nothing of the oracle is on the input side.
# ---------------------------------------------------------------------------------------------------------------------
def _mirror_pair(dim, lg, b, basis, p, nq, vname, dist, groups=None):
    """(product handler, product variant, oracle handler, oracle variant) on the same (possibly distorted) grid."""
    import polydeal_amd as pa

    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, lg)
    if dist:
        grid.distort(dist, seed=5)
    og = po.hyper_cube_refined(dim, 0.0, 1.0, lg)
    for c in range(og.n_cells):
        og.vertices[c] = grid.cell_vertices(c)
    ah, oah = pa.AgglomerationHandler(grid), po.AgglomerationHandler(og)
    if groups is None:
        ah.define_block_agglomerates(b)
        for g in po.block_agglomerates(og, b):
            oah.define_agglomerate(g)
    else:
        for g in groups:
            ah.define_agglomerate(g)
            oah.define_agglomerate(g)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(dim, p)
    ofe = (po.FE_DGQ if basis == "dgq" else po.FE_AggloDGP)(dim, p)
    for h, f in ((ah, fe), (oah, ofe)):
        h.initialize_fe_values(nq, nq)
        h.distribute_agglomerated_dofs(f)
    pvar = {"test": pa.SipVariant.minimal_sip_test, "adm": pa.SipVariant.assemble_dg_matrix,
            "poisson": lambda: pa.SipVariant.poisson_example(fe), "dr": lambda: pa.SipVariant.diffusion_reaction(fe),
            "minsip": pa.SipVariant.minimal_sip_example}[vname]()
    return grid, ah, fe, pvar, oah, variant(vname, ofe)


MIRROR_CASES = [
    (2, 3, 2, "dgq", 1, 3, "minsip", 0.0), (2, 3, 2, "dgp", 2, 3, "poisson", 0.2), (2, 3, 4, "dgq", 3, 4, "adm", 0.1),
    (3, 2, 2, "dgq", 1, 2, "test", 0.0), (3, 2, 2, "dgp", 3, 4, "poisson", 0.15), (3, 2, 2, "dgq", 2, 3, "dr", 0.1),
    (3, 2, 2, "dgq", 3, 4, "poisson", 0.0), (3, 2, 2, "dgq", 3, 4, "dr", 0.2), (3, 3, 2, "dgq", 2, 3, "adm", 0.0),
]


@pytest.mark.parametrize("case", MIRROR_CASES, ids=lambda c: "%dD_n%d_b%d_%s%d_%s_d%g" % (c[0], 2 ** c[1], c[2], c[3], c[4], c[6], c[7]))
@pytest.mark.parametrize("diag_first", [True, False])
def test_product_mirror_to_gpu_against_oracle(case, diag_first):
    import polydeal_amd as pa

    grid, ah, fe, pvar, oah, ovar = _mirror_pair(*case)
    rp, ci, vals = pa.assemble_dg_matrix(fe, ah, pvar, diag_first=diag_first)
    orp, oci, ref = po.assemble_csr(oah, ovar, diag_first=diag_first)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    # per block-scale as well as against the global maximum
    assert_parity(vals, ref, orp, oci, fe.n_dofs_per_cell)


def _rows_dense(rowptr, cols, vals, n_cols):
    out = np.zeros((len(rowptr) - 1, n_cols))
    for r in range(len(rowptr) - 1):
        out[r, cols[rowptr[r]:rowptr[r + 1]]] = vals[rowptr[r]:rowptr[r + 1]]
    return out


@pytest.mark.parametrize("basis,p,vname,dist,world", [("dgq", 2, "dr", 0.1, 3), ("dgq", 3, "poisson", 0.0, 2),
                                                      ("dgp", 2, "adm", 0.15, 4)])
def test_rank_local_descriptions_reproduce_the_global_rows(basis, p, vname, dist, world):
    """Every 'rank' gets ONLY its local + ghost description (pdh_problem.local = 1: owned and ghost polytopes, global dof
    numbers, rowptr/colind of its own rows - what an MPI rank of the reference holds, source/agglomeration_handler.cc:
    1026-1091) and reproduces its rows of the global assembly; also in the Epetra local-column order of
    TrilinosWrappers::SparseMatrix (ghost columns behind the owned ones)."""
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    grid, ah, fe, pvar, oah, ovar = _mirror_pair(3, 3 if p < 3 else 2, 2, basis, p, p + 1, vname, dist)
    n, nA, N = fe.n_dofs_per_cell, ah.n_agglomerates, ah.n_dofs
    _, _, ref = po.assemble_csr(oah, ovar, diag_first=False)
    orp, oci = oah.sparsity_pattern(False)
    dense_ref = _rows_dense(orp, oci, ref, N)
    splits = [row_range(nA, n, r, world)[0] for r in range(world)] + [N]
    for diag_first in (True, False):
        gflat = ah.flatten(pvar, diag_first, True)
        ctx = pa.Context(0)
        ctx.set_problem(gflat)
        gvals = ctx.assemble()
        ctx.close()
        grp = gflat.arrays()["rowptr"]
        for r in range(world):
            r0, r1 = splits[r], splits[r + 1]
            loc = ah.flatten_local(pvar, r0, r1, diag_first, True, row_splits=splits)
            assert loc.c.n_agg <= nA and loc.c.local == 1
            ctx = pa.Context(0)
            ctx.set_problem(loc, r0, r1)
            v = ctx.assemble()
            ctx.close()
            want = gvals[grp[r0]:grp[r1]]
            assert v.shape == want.shape
            # blocks of cut faces are contracted from the other side (and faces may be visited in another order): rounding
            assert np.max(np.abs(v - want)) <= 1e-13 * np.max(np.abs(gvals))
    # Epetra order: rows sorted by local column id
    for r in range(world):
        r0, r1 = splits[r], splits[r + 1]
        loc = ah.flatten_local(pvar, r0, r1, False, True, row_splits=splits, epetra_columns=True)
        la = loc.arrays()
        ctx = pa.Context(0)
        ctx.set_problem(loc, r0, r1)
        v = ctx.assemble()
        ctx.close()
        # local column id -> global column
        n_loc_cols = int(la["col_offset"].max()) + n
        l2g = np.zeros(n_loc_cols, dtype=np.int64)
        for a in range(loc.c.n_agg):
            l2g[la["col_offset"][a]:la["col_offset"][a] + n] = la["dof_offset"][a] + np.arange(n)
        rp_l, ci_l = la["rowptr"], la["colind"]
        assert all(np.all(np.diff(ci_l[rp_l[i]:rp_l[i + 1]]) > 0) for i in range(r1 - r0))  # rows ascending in LOCAL ids
        got = _rows_dense(rp_l, l2g[ci_l], v, N)
        assert np.max(np.abs(got - dense_ref[r0:r1])) <= TOL * np.max(np.abs(ref))
        if r > 0:  # a ghost block with a smaller global number sits behind the owned blocks
            assert any(np.any(np.diff(l2g[ci_l[rp_l[i]:rp_l[i + 1]]]) < 0) for i in range(r1 - r0))


@pytest.mark.parametrize("basis,p,vname,dist,world,alg", [("dgq", 2, "dr", 0.1, 3, "auto"), ("dgq", 3, "poisson", 0.0, 2, "moment"),
                                                          ("dgq", 3, "adm", 0.1, 4, "direct"), ("dgp", 2, "poisson", 0.15, 4, "auto")])
def test_ghost_block_exchange_equals_owner_computes_rows(basis, p, vname, dist, world, alg):
    """The reference's distributed scheme (include/poly_utils.h:1930-1992, 2134-2194: the owner of a cut face assembles M11,
    M12, M21, M22 and ships M21 / M22) as a selectable variant: W contexts on one device play the ranks, the transport is a
    device-to-device copy here (bench.py uses RCCL all-to-all).  Rows must equal owner-computes-rows and the oracle."""
    import ctypes as C
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    hip = C.CDLL("libamdhip64.so")  # the runtime the library itself is linked to (torch is not needed for device buffers)

    class DevBuf:
        def __init__(self, n_doubles, fill):
            self.n, self.p = max(n_doubles, 1), C.c_void_p()
            assert hip.hipMalloc(C.byref(self.p), C.c_size_t(8 * self.n)) == 0
            self.put(np.full(self.n, fill))

        def put(self, a, off=0):
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert hip.hipMemcpy(C.c_void_p(self.p.value + 8 * off), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0

        def get(self):
            out = np.empty(self.n)
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, C.c_size_t(out.nbytes), 2) == 0
            return out

        def data_ptr(self):
            return self.p.value

        def __del__(self):
            hip.hipFree(self.p)

    grid, ah, fe, pvar, oah, ovar = _mirror_pair(3, 3 if p < 3 else 2, 2, basis, p, p + 1, vname, dist)
    n, nA, N = fe.n_dofs_per_cell, ah.n_agglomerates, ah.n_dofs
    orp, oci, ref = po.assemble_csr(oah, ovar, diag_first=True)
    splits = [row_range(nA, n, r, world)[0] for r in range(world)] + [N]
    for diag_first in (True, False):
        if not diag_first:
            orp, oci, ref = po.assemble_csr(oah, ovar, diag_first=False)
        ctxs, sends, recvs, lay = [], [], [], []
        for r in range(world):
            loc = ah.flatten_local(pvar, splits[r], splits[r + 1], diag_first, False, row_splits=splits)
            c = pa.Context(0)
            c.set_algorithm(alg)
            c.set_exchange_mode("ghost")
            c.set_problem(loc, splits[r], splits[r + 1])
            sc, rc = c.exchange_layout(world)
            lay.append((sc, rc))
            sends.append(DevBuf(sum(sc), 0.0))
            recvs.append(DevBuf(sum(rc), np.nan))
            ctxs.append(c)
        n_cut_blocks = sum(sum(l[0]) for l in lay) // (n * n)
        assert n_cut_blocks > 0
        for r in range(world):
            for s_ in range(world):
                assert lay[r][0][s_] == lay[s_][1][r]  # what r sends to s is what s expects from r
        for r, c in enumerate(ctxs):
            c.assemble_device()
            c.exchange_get_send(sends[r].data_ptr())
            c.synchronize()
        for r in range(world):  # the transport: segment (r -> s) of r's send buffer becomes segment (from r) of s's recv buffer
            so = np.concatenate([[0], np.cumsum(lay[r][0])]).astype(np.int64)
            hs = sends[r].get()
            for s_ in range(world):
                ro = np.concatenate([[0], np.cumsum(lay[s_][1])]).astype(np.int64)
                if lay[r][0][s_]:
                    recvs[s_].put(hs[so[s_]:so[s_ + 1]], off=int(ro[r]))
        got = []
        for r, c in enumerate(ctxs):
            c.exchange_apply(recvs[r].data_ptr())
            got.append(c.values())
            c.close()
        got = np.concatenate(got)
        assert got.shape == ref.shape
        assert_parity_ah(got, ref, oah, diag_first, what="ghost exchange")
        # against owner-computes-rows on the same descriptions
        own = []
        for r in range(world):
            loc = ah.flatten_local(pvar, splits[r], splits[r + 1], diag_first, False, row_splits=splits)
            c = pa.Context(0)
            c.set_algorithm(alg)
            c.set_problem(loc, splits[r], splits[r + 1])
            own.append(c.assemble())
            c.close()
        assert np.max(np.abs(got - np.concatenate(own))) <= 1e-13 * np.max(np.abs(ref))



# ---------------------------------------------------------------------------------------------------------------------
# Row kernel (csrc/pdh_rows.h): FE_DGQ(3) on polytopes with axis-aligned planar faces; one wave writes all blocks of a
# polytope's rows (Kronecker form of the coupling blocks, rank-one face moments, whole-line stores).
# ---------------------------------------------------------------------------------------------------------------------
def _staircase_groups(grid):
    """Irregular agglomerates of Cartesian cells (L-shapes, a 1x1x2 bar, singletons): every polytopal face run is still
    planar?  No - an L-shape touches a neighbour along two planes.  Used to check that such problems FALL BACK."""
    c = lambda i, j, k: int(grid.ijk_to_cell[(i, j, k)])
    groups = [sorted([c(0, 0, 0), c(1, 0, 0), c(0, 1, 0)]), sorted([c(1, 1, 0), c(1, 1, 1)])]
    used = {x for g in groups for x in g}
    groups += [[x] for x in range(grid.n_cells) if x not in used]
    return groups


def _box_groups(grid, n):
    """Axis-aligned boxes of different sizes tiling an n^3 grid: a 2x2x2 box, 1x1x2 bars, 2x1x1 bars, singletons - all
    faces planar, neighbours of different size (frames differ: second record pass, per-face tables)."""
    c = lambda i, j, k: int(grid.ijk_to_cell[(i, j, k)])
    groups, used = [], set()

    def box(i0, j0, k0, di, dj, dk):
        g = sorted(c(i, j, k) for i in range(i0, i0 + di) for j in range(j0, j0 + dj) for k in range(k0, k0 + dk))
        assert not (set(g) & used)
        used.update(g)
        groups.append(g)

    box(0, 0, 0, 2, 2, 2)
    box(2, 0, 0, 2, 1, 2)  # +x block split along y
    box(2, 1, 0, 2, 1, 2)
    box(0, 2, 0, 1, 2, 2)  # +y block split along x
    box(1, 2, 0, 1, 2, 2)
    box(0, 0, 2, 2, 1, 2)  # +z block split along y
    box(0, 1, 2, 2, 1, 2)
    box(2, 2, 0, 2, 2, 1)  # split along z
    box(2, 2, 1, 2, 2, 1)
    box(2, 0, 2, 2, 2, 2)
    box(0, 2, 2, 2, 2, 2)
    box(2, 2, 2, 1, 2, 2)
    box(3, 2, 2, 1, 2, 2)
    assert len(used) == grid.n_cells
    # a planar-faced but non-convex set would still need one plane per neighbour: boxes guarantee it
    return groups


ROWS_CASES = [
    # lg, block (0: mixed boxes), variant, diag_first, world
    (2, 2, "poisson", True, 1), (2, 2, "dr", False, 1), (3, 2, "adm", True, 3), (2, 1, "test", True, 1),
    (2, 0, "poisson", True, 1), (2, 0, "dr", False, 2), (1, 2, "poisson", True, 1),
    (3, 4, "poisson", True, 1),  # 64 cells per polytope, 16 sub-faces per face: several batches of cells / lane tasks
]


@pytest.mark.parametrize("basis,p", [("dgq", 3), ("dgp", 3), ("dgq", 2), ("dgp", 2), ("dgq", 1), ("dgp", 1)])
@pytest.mark.parametrize("lg,b,vname,diag_first,world", ROWS_CASES)
def test_row_kernel_parity(lg, b, vname, diag_first, world, basis, p):
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    nq = p + 1
    grid = po.hyper_cube_refined(3, 0.0, 1.0, lg)
    ah = po.AgglomerationHandler(grid)
    for g in (po.block_agglomerates(grid, b) if b else _box_groups(grid, 4)):
        ah.define_agglomerate(g)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = variant(vname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    sc = np.max(np.abs(ref))
    has_general_paths = p == 3  # the kinds of lower degree exist for tensor rules only
    # no claim about the rules (0): the library finds their tensor structure on the points
    v0, used_0, kern_0 = _values_k(kw, "rows")
    assert used_0 == "rows"
    assert_parity_ah(v0, ref, ah, diag_first, what="rows")
    # every element takes a term kernel (pdh_terms.h; FE_DGQ(3): its workgroup form, pdh_terms_wg.h) while a polytope's tables fit
    # the LDS budget, else the kinds of pdh_rows.h - which must agree with it to rounding
    # (blocks of 4^3 cells: 64 cells and 96 sub-faces would not fit, but they form tensor grids and are summed over as 8 cells and 24
    # sub-faces with composite rules - pdh_terms_merge_stats; the box-shaped agglomerates of the b = 0 cases likewise)
    assert kern_0 == "terms", kern_0
    vs, used_s, kern_s = _values_k(kw, "rows", terms=False)
    assert used_s == "rows" and kern_s == ("pieces" if (p == 3 and basis == "dgq") else "streamed")
    assert_parity_ah(vs, ref, ah, diag_first, what="rows (kinds of pdh_rows.h)")
    assert np.max(np.abs(vs - v0)) <= 1e-13 * sc
    # verified claims: bit-identical
    vf, used_f = _values(dict(kw, fq_tensor_n=nq, vq_tensor_n=nq), "rows")
    assert used_f == "rows" and np.array_equal(v0, vf)
    if has_general_paths:
        # general-point paths (tensor structure of the rules neither claimed nor looked for)
        vr, used = _values(dict(kw, vq_tensor_n=-1, fq_tensor_n=-1), "rows")
        assert used == "rows"
        assert_parity_ah(vr, ref, ah, diag_first, what="rows, general points")
        assert np.max(np.abs(v0 - vr)) > 0.0  # really another path
        # each structure alone
        for hint in (dict(vq_tensor_n=nq, fq_tensor_n=-1), dict(fq_tensor_n=nq, vq_tensor_n=-1)):
            vh, used_h = _values(dict(kw, **hint), "rows")
            assert used_h == "rows"
            assert_parity_ah(vh, ref, ah, diag_first, what=str(hint))
            assert np.max(np.abs(vh - vr)) > 0.0 and np.max(np.abs(vh - v0)) > 0.0
        # a wrong claim must be harmless (the check on the points fails, the general path is taken)
        vw, _ = _values(dict(kw, vq_tensor_n=2, fq_tensor_n=2), "rows")
        assert np.array_equal(vw, vr)
    else:
        # without the tensor structure these kinds must refuse (forced) / fall back (AUTO)
        va, used_a = _values(dict(kw, vq_tensor_n=-1, fq_tensor_n=-1), "auto")
        assert used_a != "rows"
        assert_parity_ah(va, ref, ah, diag_first)
        prob = pa.Problem(**dict(kw, vq_tensor_n=-1, fq_tensor_n=-1))
        ctx = pa.Context(0)
        ctx.set_problem(prob)
        ctx.set_algorithm("rows")
        with pytest.raises(pa.PdhError):
            ctx.assemble()
        ctx.close()
    other = "moment" if (basis == "dgq" and p == 3) else "direct"
    vm, used_m = _values(kw, other)
    assert used_m == other and np.max(np.abs(v0 - vm)) <= 1e-13 * sc
    n = fe.n_dofs_per_cell
    if world > 1:
        parts = []
        for r in range(world):
            rb, re = row_range(ah.n_agglomerates, n, r, world)
            v, u = _values(kw, "rows", rb, re)
            assert u == "rows"
            parts.append(v)
        assert_parity_ah(np.concatenate(parts), ref, ah, diag_first, what="row ranges")


def test_row_kernel_falls_back_when_faces_are_not_planar():
    """AUTO must not pick the row kernel on distorted grids (forcing it fails).  A polytopal face that spans two planes of a
    Cartesian grid is no reason to fall back: FE_DGQ(3) has the MULTI instantiation of pdh_rows.h, the other elements the term
    kernel (pdh_terms.h)."""
    import polydeal_amd as pa

    for mode, fe, expect in (("distorted", po.FE_DGQ(3, 3), "moment"), ("staircase", po.FE_DGQ(3, 3), "rows"),
                             ("staircase", po.FE_DGQ(3, 2), "rows"), ("staircase", po.FE_AggloDGP(3, 3), "rows"),
                             ("distorted", po.FE_DGQ(3, 2), "mixed"), ("distorted", po.FE_AggloDGP(3, 3), "direct")):
        grid = po.hyper_cube_refined(3, 0.0, 1.0, 2)
        if mode == "distorted":
            grid.distort(1e-9, seed=1)  # even a tiny perturbation: the kernel evaluates ONE plane coordinate per face
        ah = po.AgglomerationHandler(grid)
        for g in (po.block_agglomerates(grid, 2) if mode == "distorted" else _staircase_groups(grid)):
            ah.define_agglomerate(g)
        nq = fe.degree + 1
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        var = po.variant_poisson_example(fe)
        kw = flatten(ah, var)
        ref = po.assemble_csr(ah, var)[2]
        va, used = _values(kw, "auto")
        assert used == expect, (mode, fe.name, used)
        assert_parity_ah(va, ref, ah, what="%s %s" % (mode, fe.name))
        if expect != "rows":
            prob = pa.Problem(**kw)
            ctx = pa.Context(0)
            ctx.set_problem(prob)
            ctx.set_algorithm("rows")
            with pytest.raises(pa.PdhError):
                ctx.assemble()
            ctx.close()


def _grown_agglomerates(grid, cells_per_polytope, seed):
    """METIS stand-in (METIS is not available offline; reference examples/poisson.cc:543-566 partitions the cell connectivity
    graph into connected parts of about equal size): regions grown over the cell adjacency graph from random seeds."""
    rng = np.random.default_rng(seed)
    return random_agglomeration(grid, max(2, grid.n_cells // cells_per_polytope), rng)


@pytest.mark.parametrize("cells,per,vname,diag_first,seed", [(4, 4, "poisson", True, 0), (4, 8, "dr", False, 1), (6, 6, "adm", True, 2),
                                                             (6, 3, "test", True, 3), (8, 8, "poisson", True, 4)])
def test_row_kernel_staircase_agglomerates(cells, per, vname, diag_first, seed, monkeypatch):
    """Irregular agglomerates of Cartesian cells (regions grown over the cell graph, the METIS stand-in): neighbours are met
    along several planes, polytopes have many more than six faces, boundary runs span up to five planes.  FE_DGQ(3) takes the row
    kernel (MULTI instantiation: a coupling block is the sum of the Kronecker products of its planes): parity with the oracle
    per block, both CSR layouts, with and without the tensor structure of the rules, on row ranges; agreement with the moment
    form to rounding.  (PDH_TERMS_DGQ3=0: AUTO would take the workgroup term kernel, test_term_kernel_workgroup_form_for_dgq3.)"""
    from polydeal_amd.partition import row_range

    monkeypatch.setenv("PDH_TERMS_DGQ3", "0")
    fe = po.FE_DGQ(3, 3)
    grid = po.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    groups = _grown_agglomerates(grid, per, seed)
    ah = po.AgglomerationHandler(grid)
    order = np.random.default_rng(seed).permutation(len(groups))  # polytope index order != master cell order
    for k in order:
        ah.define_agglomerate(groups[k])
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    assert max(ah.n_faces) > 6  # more neighbours than a box has
    var = variant(vname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    v0, used = _values(kw, "auto")
    assert used == "rows"
    assert_parity_ah(v0, ref, ah, diag_first, what="rows (tensor rules found)")
    vg, used = _values(dict(kw, vq_tensor_n=-1, fq_tensor_n=-1), "rows")
    assert used == "rows"
    assert_parity_ah(vg, ref, ah, diag_first, what="rows (general points)")
    assert np.max(np.abs(vg - v0)) > 0.0  # really the other path
    vm, used = _values(kw, "moment")
    assert used == "moment"
    assert np.max(np.abs(v0 - vm)) <= 1e-13 * np.max(np.abs(ref))
    n = fe.n_dofs_per_cell
    parts = []
    for r in range(3):
        rb, re = row_range(ah.n_agglomerates, n, r, 3)
        v, u = _values(kw, "rows", rb, re)
        assert u == "rows"
        parts.append(v)
    assert_parity_ah(np.concatenate(parts), ref, ah, diag_first, what="row ranges")


@pytest.mark.parametrize("basis,p", [("dgp", 3), ("dgq", 2), ("dgp", 2), ("dgq", 1), ("dgp", 1)])
@pytest.mark.parametrize("cells,per,vname,diag_first,seed", [(4, 4, "poisson", True, 0), (4, 8, "dr", False, 1), (6, 6, "adm", True, 2),
                                                             (6, 3, "minsip", True, 3), (8, 8, "poisson", False, 4)])
def test_term_kernel_staircase_agglomerates(cells, per, vname, diag_first, seed, basis, p):
    """The element the reference's own callers instantiate on the agglomerates they produce (examples/poisson.cc:413 FE_AggloDGP,
    :543-566 METIS; here regions grown over the cell graph): neighbours met along several planes, many more than six faces,
    boundary runs over up to five planes.  AUTO takes the row algorithm through the term kernel (pdh_terms.h) for all five small
    elements: parity with the oracle per block, both CSR layouts, every caller variant (zeroed boundary: `minsip`), row ranges,
    agreement with the two-kernel forms to rounding."""
    from polydeal_amd.partition import row_range

    fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    grid = po.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    groups = _grown_agglomerates(grid, per, seed)
    ah = po.AgglomerationHandler(grid)
    order = np.random.default_rng(seed).permutation(len(groups))  # polytope index order != master cell order
    for k in order:
        ah.define_agglomerate(groups[k])
    nq = p + 1
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    assert max(ah.n_faces) > 6  # more neighbours than a box has
    var = variant(vname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    v0, used, kern = _values_k(kw, "auto")
    assert used == "rows" and kern == "terms"
    assert_parity_ah(v0, ref, ah, diag_first, what="term kernel")
    # without it these meshes take the two-kernel forms (the streamed kinds of pdh_rows.h need one plane per neighbour)
    vd, used_d, kern_d = _values_k(kw, "auto", terms=False)
    assert used_d != "rows" and kern_d == "none"
    assert np.max(np.abs(v0 - vd)) <= 1e-13 * np.max(np.abs(ref))
    n = fe.n_dofs_per_cell
    parts = []
    for r in range(3):
        rb, re = row_range(ah.n_agglomerates, n, r, 3)
        v, u, k = _values_k(kw, "rows", rb, re)
        assert u == "rows" and k == "terms"
        parts.append(v)
    assert_parity_ah(np.concatenate(parts), ref, ah, diag_first, what="term kernel, row ranges")


@pytest.mark.parametrize("basis,p,nq", [("dgp", 3, 7), ("dgq", 2, 5), ("dgp", 2, 6), ("dgq", 1, 3), ("dgp", 1, 8)])
def test_term_kernel_rules_of_up_to_eight_points(basis, p, nq):
    """QGauss(2p+1) and other rules with more points per direction than p + 1 (examples/minimal_SIP.cc:151-157; the tests of the
    reference): the term kernel's instantiation for up to 8 points per direction, on boxes of different sizes (neighbours with
    other bounding boxes) and on blocks."""
    fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    for b in (0, 2):
        grid = po.hyper_cube_refined(3, 0.0, 1.0, 2)
        ah = po.AgglomerationHandler(grid)
        for g in (po.block_agglomerates(grid, b) if b else _box_groups(grid, 4)):
            ah.define_agglomerate(g)
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        var = po.variant_assemble_dg_matrix()
        kw = flatten(ah, var)
        ref = po.assemble_csr(ah, var)[2]
        v0, used, kern = _values_k(kw, "auto")
        assert used == "rows" and kern == "terms"
        assert_parity_ah(v0, ref, ah, what="term kernel, %d points per direction" % nq)


def _merge_stats_and_values(kw, mode):
    import polydeal_amd as pa

    old = os.environ.get("PDH_TERMS_MERGE")
    os.environ["PDH_TERMS_MERGE"] = mode
    try:
        ctx = pa.Context(0)
        ctx.set_problem(pa.Problem(**kw))
        kern, st = ctx.rows_kernel_in_use(), ctx.terms_merge_stats()
        v = ctx.assemble()
        ctx.close()
    finally:
        if old is None:
            del os.environ["PDH_TERMS_MERGE"]
        else:
            os.environ["PDH_TERMS_MERGE"] = old
    return v, kern, st


@pytest.mark.parametrize("basis,p", [("dgq", 3), ("dgp", 3), ("dgq", 2), ("dgp", 1)])
@pytest.mark.parametrize("shape,diag_first,vname", [("blocks3", True, "poisson"), ("blocks3", False, "dr"), ("slabs", True, "adm"),
                                                   ("grown", True, "poisson"), ("grown", False, "test"), ("graded", True, "poisson"),
                                                   ("graded", False, "dr")])
def test_term_kernels_merge_cells_and_sub_faces_that_form_tensor_grids(basis, p, shape, diag_first, vname):
    """The term kernels sum over cells and sub-faces; those that form tensor grids are merged into ONE with composite 1-D rules
    (csrc/pdh_capi.cpp: merge_terms_of_slot).  Shapes that exercise the analysis: blocks of 3^3 cells (three intervals per axis: sub-grids of
    2 + 1), slabs of 4 x 2 x 1 cells (different counts per axis, composite rules of 8 and 4 and 2 points... per the element's rule), and
    METIS-like grown agglomerates with PDH_TERMS_MERGE=2 (every polytope a mix of merged sub-grids, merged planes and single cells /
    sub-faces - the rule of build_terms_tables would leave them as given), and 2^3-cell blocks of a GRADED Cartesian grid (vertices
    x -> x^1.6 per axis: the intervals of a composite rule differ in length, the weights of a polytope's cells still factorise).
    Merged, as given and the oracle must agree."""
    fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    nq = p + 1
    if shape == "blocks3":
        grid = po.subdivided_hyper_cube(3, 6, 0.0, 1.0)
        groups = po.block_agglomerates(grid, 3)
    elif shape == "slabs":
        grid = po.subdivided_hyper_cube(3, 4, 0.0, 1.0)
        groups = {}
        for i, j, k in np.ndindex(*grid.ijk_to_cell.shape):
            groups.setdefault((i // 4, j // 2, k), []).append(int(grid.ijk_to_cell[i, j, k]))
        groups = [sorted(g) for _, g in sorted(groups.items())]
    elif shape == "graded":
        grid = po.subdivided_hyper_cube(3, 4, 0.0, 1.0)
        grid.vertices[...] = grid.vertices ** 1.6  # (a tensor-product grading: cells stay axis-aligned boxes)
        groups = po.block_agglomerates(grid, 2)
    else:
        grid = po.subdivided_hyper_cube(3, 6, 0.0, 1.0)
        groups = _grown_agglomerates(grid, 6, 5)
    ah = po.AgglomerationHandler(grid)
    for g in groups:
        ah.define_agglomerate(g)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = variant(vname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    vm, kern_m, st_m = _merge_stats_and_values(kw, "2")
    vg, kern_g, st_g = _merge_stats_and_values(kw, "0")
    assert kern_m == "terms" and kern_g == "terms"
    assert st_g["cells_merged"] == st_g["cells"] and st_g["sub_faces_merged"] == st_g["sub_faces"]
    assert st_m["cells"] == st_g["cells"] == grid.n_cells and st_m["sub_faces"] == st_g["sub_faces"]
    m = min(4, 8 // nq)       # intervals of a composite rule: 8 points, 4 intervals at most
    up = lambda a: -(-a // m)
    if shape == "blocks3":    # 27 cells -> sub-grids of m intervals per axis; a face of 9 sub-faces likewise
        assert st_m["cells_merged"] * 27 == st_m["cells"] * up(3) ** 3 and st_m["sub_faces_merged"] * 9 == st_m["sub_faces"] * up(3) ** 2
    elif shape == "slabs":    # 4 x 2 x 1 cells
        assert st_m["cells_merged"] * 8 == st_m["cells"] * up(4) * up(2)
    elif shape == "graded":   # 2^3 cells of unequal size -> one
        assert st_m["cells_merged"] * 8 == st_m["cells"] and st_m["sub_faces_merged"] * 4 == st_m["sub_faces"]
    else:
        assert st_m["sub_faces_merged"] < st_m["sub_faces"]
    assert_parity_ah(vm, ref, ah, diag_first, what="term kernel, merged")
    assert_parity_ah(vg, ref, ah, diag_first, what="term kernel, as given")
    assert np.max(np.abs(vm - vg)) <= 1e-13 * np.max(np.abs(ref))


@pytest.mark.parametrize("waves", ["4", "8"])
@pytest.mark.parametrize("kind,cells,per,vname,diag_first,nq", [("block", 4, 2, "poisson", True, 4), ("block", 6, 2, "dr", False, 4), ("block", 4, 1, "test", True, 4),
                                                                ("grown", 6, 6, "adm", True, 4), ("grown", 8, 8, "poisson", False, 4), ("grown", 6, 3, "minsip", True, 4),
                                                                ("block", 4, 2, "adm", True, 7), ("boxes", 4, 0, "poisson", True, 4)])
def test_term_kernel_workgroup_form_for_dgq3(kind, cells, per, vname, diag_first, nq, waves, monkeypatch):
    """FE_DGQ(3) through the workgroup-per-polytope form of the term kernel (pdh_terms_wg.h: what AUTO takes for that element; four or
    eight waves per polytope): blocks, boxes of different sizes, staircase agglomerates, both CSR layouts, rules of 4 and 7 points
    per direction, row ranges - per-block parity with the oracle and agreement with the kinds of pdh_rows.h to rounding."""
    import subprocess
    import sys

    # (the number of waves is read once per process by the launcher: one child process per value)
    if waves != "4":
        here = __import__('os').path.dirname(__import__('os').path.abspath(__file__))
        code = ("import os, sys; os.environ['PDH_TERMS_WG_WAVES'] = %r; sys.path.insert(0, %r); sys.path.insert(0, %r); "
                "import test_gpu_parity as t; t._wg_case(%r, %r, %r, %r, %r, %r)"
                % (waves, __import__('os').path.dirname(here), here, kind, cells, per, vname, diag_first, nq))
        env = dict(__import__('os').environ, PDH_TERMS_DGQ3="1", PDH_TERMS_WG_WAVES=waves)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        return
    monkeypatch.setenv("PDH_TERMS_DGQ3", "1")
    _wg_case(kind, cells, per, vname, diag_first, nq)


def _wg_case(kind, cells, per, vname, diag_first, nq):
    import os

    from polydeal_amd.partition import row_range

    os.environ["PDH_TERMS_DGQ3"] = "1"
    fe = po.FE_DGQ(3, 3)
    grid = po.subdivided_hyper_cube(3, cells, 0.0, 1.0) if kind != "boxes" else po.hyper_cube_refined(3, 0.0, 1.0, 2)
    if kind == "block":
        groups = po.block_agglomerates(grid, per)
    elif kind == "boxes":
        groups = _box_groups(grid, 4)
    else:
        groups = _grown_agglomerates(grid, per, cells)
    ah = po.AgglomerationHandler(grid)
    for g in groups:
        ah.define_agglomerate(g)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = variant(vname, fe)
    kw = flatten(ah, var, diag_first=diag_first)
    ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
    v0, used, kern = _values_k(kw, "auto")
    assert used == "rows" and kern == "terms"
    assert_parity_ah(v0, ref, ah, diag_first, what="workgroup term kernel")
    os.environ["PDH_TERMS_DGQ3"] = "0"
    try:
        v1, used1, kern1 = _values_k(kw, "auto")
    finally:
        os.environ["PDH_TERMS_DGQ3"] = "1"
    assert used1 == "rows" and kern1 in ("pieces", "multi")
    assert np.max(np.abs(v0 - v1)) <= 1e-13 * np.max(np.abs(ref))
    n = fe.n_dofs_per_cell
    parts = []
    for r in range(2):
        rb, re = row_range(ah.n_agglomerates, n, r, 2)
        v, u, k = _values_k(kw, "rows", rb, re)
        assert u == "rows" and k == "terms"
        parts.append(v)
    assert_parity_ah(np.concatenate(parts), ref, ah, diag_first, what="workgroup term kernel, row ranges")


def test_row_kernel_staircase_with_large_plane_entries(monkeypatch):
    """MULTI instantiation with plane entries of MORE sub-faces than one staging round of its S takes (14): 4x4x4 blocks of an
    8^3 grid, one corner cell moved from block A to its +x neighbour B - A and B then meet along a 15-sub-face plane, a
    16-sub-face plane's worth of neighbours elsewhere, and the moved cell adds one-sub-face planes of three axes.  Parity with
    the oracle per block; tensor rules and general points.  (PDH_TERMS_DGQ3=0: pdh_rows.h, not the workgroup term kernel.)"""
    monkeypatch.setenv("PDH_TERMS_DGQ3", "0")
    fe = po.FE_DGQ(3, 3)
    grid = po.subdivided_hyper_cube(3, 8, 0.0, 1.0)
    c = lambda i, j, k: int(grid.ijk_to_cell[(i, j, k)])
    blocks = {}
    for i in range(8):
        for j in range(8):
            for k in range(8):
                blocks.setdefault((i // 4, j // 4, k // 4), []).append(c(i, j, k))
    moved = c(3, 0, 0)  # corner cell of block (0,0,0) on its +x face
    blocks[(0, 0, 0)].remove(moved)
    blocks[(1, 0, 0)].append(moved)
    ah = po.AgglomerationHandler(grid)
    for key in sorted(blocks):
        ah.define_agglomerate(sorted(blocks[key]))
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    for diag_first in (True, False):
        kw = flatten(ah, var, diag_first=diag_first)
        ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
        v0, used = _values(kw, "auto")
        assert used == "rows"
        assert_parity_ah(v0, ref, ah, diag_first, what="rows, large plane entries (tensor rules)")
        vg, used = _values(dict(kw, vq_tensor_n=-1, fq_tensor_n=-1), "rows")
        assert used == "rows"
        assert_parity_ah(vg, ref, ah, diag_first, what="rows, large plane entries (general points)")


def test_poisson_sanity_check_02_on_gpu():
    """test/polydeal/poisson_sanity_check_02.output ('Step function = 2', 'V function = 1') with the matrix assembled by the
    product chain (host mirror -> C ABI -> HIP kernels)."""
    import polydeal_amd as pa

    grid = pa.BackgroundGrid.hyper_cube_refined(2, 0.0, 1.0, 1)
    ah = pa.AgglomerationHandler(grid)
    ah.define_agglomerate([0, 2])
    ah.define_agglomerate([1, 3])
    fe = pa.FE_DGQ(2, 1)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    rp, ci, vals = pa.assemble_dg_matrix(fe, ah, pa.SipVariant.minimal_sip_example(), diag_first=True)
    A = po.csr_to_dense(rp, ci, vals, ah.n_dofs)
    # nodal interpolation on the bounding boxes (VectorTools::interpolate with the agglomeration mapping)
    out = []
    for name, f in (("Step function", lambda x: (x >= 0.5) * 1.0), ("V function", lambda x: np.abs(x - 0.5))):
        v = np.zeros(ah.n_dofs)
        for P in range(ah.n_agglomerates):
            lo, hi = ah.bbox(P)
            xs = np.array([lo[0], hi[0], lo[0], hi[0]])  # FE_DGQ(1) support points, x fastest
            v[ah.dof_indices(P)] = f(xs)
        out.append("Test with %s = %s" % (name, gc.fmt(float(v @ A @ v))))
    assert out == gc.golden_lines("poisson_sanity_check_02.output")[:-1]


def test_small_problem_graph_replay_is_transparent():
    """Launch-bound problems replay their two kernels from a hipGraph (pdh_capi.cpp: graph_exec): repeated assemblies, a change of
    algorithm, a new problem on the same context and profiling (plain launches) must all give the same matrix."""
    import polydeal_amd as pa

    fe = po.FE_DGQ(2, 2)
    ah = build(2, 3, 2, fe, 3)
    var = po.variant_poisson_example(fe)
    kw = flatten(ah, var)
    ref = po.assemble_csr(ah, var)[2]
    sc = np.max(np.abs(ref))
    ctx = pa.Context(0)
    ctx.set_problem(pa.Problem(**kw))
    first = ctx.assemble()          # captures
    assert_parity_ah(first, ref, ah)
    for _ in range(3):              # replays
        assert np.array_equal(ctx.assemble(), first)
    ctx.set_profiling(True)         # plain launches with events
    assert np.array_equal(ctx.assemble(), first)
    ctx.set_profiling(False)
    assert np.array_equal(ctx.assemble(), first)
    # another problem on the same context: the old graph must be gone
    fe2 = po.FE_AggloDGP(2, 3)
    ah2 = build(2, 2, 2, fe2, 4, distort=0.1)
    var2 = po.variant_poisson_example(fe2)
    kw2 = flatten(ah2, var2)
    ref2 = po.assemble_csr(ah2, var2)[2]
    ctx.set_problem(pa.Problem(**kw2))
    for _ in range(2):
        got = ctx.assemble()
        assert_parity_ah(got, ref2, ah2)
    # 3-D, moment form <-> direct form on the same resident problem: the graph follows the algorithm
    fe3 = po.FE_DGQ(3, 3)
    ah3 = build(3, 2, 2, fe3, 4, distort=0.1)
    var3 = po.variant_poisson_example(fe3)
    kw3 = flatten(ah3, var3)
    ref3 = po.assemble_csr(ah3, var3)[2]
    ctx.set_problem(pa.Problem(**kw3))
    for alg in ("moment", "direct", "moment", "auto"):
        ctx.set_algorithm(alg)
        for _ in range(2):
            got = ctx.assemble()
            assert_parity_ah(got, ref3, ah3, what=alg)
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own known-answer test at the HEADLINE degree (test/polydeal/minimal_SIP_Poisson.cc:486-509): the matrix of
# the agglomerated problem (2x2x2 blocks) equals, entry by entry to 1e-13, the matrix of standard SIP on the coarse mesh
# (one polytope per coarse cell).  The test is degree-generic in the reference (:78 `dg_fe(1)` is a constructor argument,
# :247-253 QGauss(2p+1) for cells and faces, :101/:308 penalty 20 with h_f = 1) and exact for any Gauss rule with >= p+1 points
# on Cartesian cells, so it pins the 3-D values of degree 2 and 3 - for which the reference holds no fixture - without the
# oracle: both matrices come out of the HIP path, through each of its three algorithms.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rule", ["2p+1", "p+1"])
@pytest.mark.parametrize("basis,p", [("dgq", 2), ("dgq", 3), ("dgp", 2), ("dgp", 3)])
def test_minimal_SIP_Poisson_identity_3d_degree_2_and_3(basis, p, rule):
    fe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    var = po.variant_minimal_sip_test()
    nq = 2 * p + 1 if rule == "2p+1" else p + 1
    n = fe.n_dofs_per_cell

    def handler(agglomerated):
        grid = po.hyper_cube_refined(3, -1.0, 1.0, 2 if agglomerated else 1)  # 64 fine cells / 8 coarse cells (Morton order)
        ah = po.AgglomerationHandler(grid)
        for k in range(8):
            ah.define_agglomerate(list(range(8 * k, 8 * k + 8)) if agglomerated else [k])  # children of coarse cell k
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        return ah

    std, agg = handler(False), handler(True)
    assert all(np.allclose(std.bbox(k), agg.bbox(k), atol=0, rtol=0) for k in range(8))
    kw_s, kw_a = flatten(std, var), flatten(agg, var)
    assert np.array_equal(kw_s["rowptr"], kw_a["rowptr"]) and np.array_equal(kw_s["colind"], kw_a["colind"])
    standard, used = _values(kw_s, "direct")
    assert used == "direct"
    scale = np.max(np.abs(standard))
    assert scale > 1.0
    seen = set()
    for alg in ("rows", "moment", "direct"):
        v, used = _values(kw_a, alg)
        assert used == alg
        seen.add(used)
        # entry_tol of the reference is absolute 1e-13 on entries of size O(10): here relative to the largest entry, and per
        # block (coupling blocks are checked at their own scale)
        assert_parity(v, standard, kw_a["rowptr"], kw_a["colind"], n, tol=1e-13, floor=1e-3, what="%s vs standard SIP" % alg)
        # the standard problem through the same algorithm as well (single-cell polytopes)
        vs, used_s = _values(kw_s, alg)
        assert used_s == alg
        assert_parity(vs, standard, kw_s["rowptr"], kw_s["colind"], n, tol=1e-13, floor=1e-3, what="standard, %s" % alg)
    assert seen == {"rows", "moment", "direct"}
    # and the oracle agrees with both (it is pinned by the reference's p = 1 fixture of this test)
    ref = po.assemble_csr(agg, var)[2]
    assert_parity(standard, ref, kw_a["rowptr"], kw_a["colind"], n, tol=1e-13, floor=1e-3, what="oracle")


def test_smoke_sizes_regression_bound():
    """Regression guard at the sizes of __graft_entry__.smoke(): the HIP path agrees with the oracle to a few ulp there
    (2.4e-15 observed); 1e-13 - globally and per block - leaves one order for summation-order changes and none for a kernel
    edit that loses digits (the 1e-12 of the north star is the acceptance bound, not the regression bound)."""
    for fe_cls in (po.FE_DGQ, po.FE_AggloDGP):
        fe = fe_cls(3, 3)
        ah = build(3, 2, 2, fe, 4)
        var = po.variant_poisson_example(fe)
        rp, ci, ref = po.assemble_csr(ah, var)
        kw = flatten(ah, var)
        for alg in ("auto", "moment", "direct"):
            v, used = _values(kw, alg)
            rel = assert_parity(v, ref, rp, ci, fe.n_dofs_per_cell, tol=1e-13, floor=1e-3, what="%s %s" % (fe.name, used))
            assert rel <= 1e-13


def test_row_store_sites_every_variant_repeatable():
    """The row stores of csrc/pdh_rows.h are inline asm (scalar row base + 32-bit lane offset), invisible to the compiler's
    hazard recogniser.  3 x 3 x 3 polytopes: the centre one has blocks left AND right of its diagonal along every axis (shifted
    pieces with carries, plain pieces, the own block's piece with a carry), the corner ones start with their own block
    (no carry) - every store site of P4 / P5, in both CSR layouts and for every kind.  A value stored before its
    producer has finished, or into a neighbouring slot, is an O(1) error in some entry; launches must also repeat bit for bit."""
    import polydeal_amd as pa

    grid = po.subdivided_hyper_cube(3, 6, 0.0, 1.0)
    for fe in (po.FE_DGQ(3, 3), po.FE_AggloDGP(3, 3), po.FE_DGQ(3, 2)):
        ah = po.AgglomerationHandler(grid)
        for g in po.block_agglomerates(grid, 2):
            ah.define_agglomerate(g)
        nq = fe.degree + 1
        ah.initialize_fe_values(nq, nq)
        ah.distribute_agglomerated_dofs(fe)
        var = po.variant_diffusion_reaction(fe)
        for diag_first in (True, False):
            kw = flatten(ah, var, diag_first=diag_first)
            ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
            ctx = pa.Context(0)
            ctx.set_algorithm("rows")
            ctx.set_problem(pa.Problem(**kw))
            assert ctx.algorithm_in_use() == "rows"
            first = ctx.assemble()
            assert_parity_ah(first, ref, ah, diag_first, tol=1e-13, floor=1e-3, what="%s diag_first=%s" % (fe.name, diag_first))
            for _ in range(8):
                ctx.poison_values()
                assert np.array_equal(ctx.assemble(), first)
            ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# Unstructured background mesh (meshes/t3.msh of the reference = tests/golden/t3.msh): the reference's own callers run on it
# (examples/minimal_SIP.cc:94-139, test/polydeal/poisson_sanity_check_03.cc:103-135).
# ---------------------------------------------------------------------------------------------------------------------
def test_poisson_sanity_check_03_on_t3_mesh_on_gpu():
    """All six runs of test/polydeal/poisson_sanity_check_03.output (N = 50 .. 800 agglomerates of t3.msh refined 3 times,
    FE_DGQ(1), QGauss(3), boundary terms dropped) with the matrix from the HIP path - product mirror (gmsh reader, refine_global,
    grown agglomerates: METIS stand-in, the identities hold for any agglomeration) -> C ABI -> kernels."""
    import os
    import scipy.sparse as sp
    import polydeal_amd as pa

    mesh = os.path.join(gc.GOLDEN_DIR, "t3.msh")
    out = []
    for n_sub in (50, 100, 120, 300, 400, 800):
        grid = pa.BackgroundGrid.read_msh(mesh, 3)
        assert grid.n_cells == 5824
        ah = pa.AgglomerationHandler(grid)
        ah.partition_into_grown_agglomerates(n_sub, seed=n_sub)
        assert ah.n_agglomerates == n_sub
        fe = pa.FE_DGQ(2, 1)
        ah.initialize_fe_values(3, 3)
        ah.distribute_agglomerated_dofs(fe)
        var = pa.SipVariant(10.0, 1, 0, 1)  # penalty 10 / h, index()<index(), boundary zeroed (:96, 230-262)
        rp, ci, vals = pa.assemble_dg_matrix(fe, ah, var)
        A = sp.csr_matrix((vals, ci, rp), shape=(ah.n_dofs, ah.n_dofs))
        arr = ah.flatten(var, True, False).arrays()
        bb = arr["bbox"].reshape(-1, 2, 2)
        corner = np.array([[0, 0], [1, 0], [0, 1], [1, 1]])  # FE_DGQ(1) nodes: corners of the box, x fastest
        x = np.zeros(ah.n_dofs)
        y = np.zeros(ah.n_dofs)
        for i in range(4):
            x[arr["dof_offset"] + i] = bb[np.arange(n_sub), corner[i, 0], 0]
            y[arr["dof_offset"] + i] = bb[np.arange(n_sub), corner[i, 1], 1]
        one = np.ones(ah.n_dofs)
        forms = [float(v @ (A @ v)) for v in (x, x + y, one)]
        assert abs(forms[0] - 1.0) < 1e-12 and abs(forms[1] - 2.0) < 1e-12 and abs(forms[2]) < 1e-12, (n_sub, forms)
        out += ["N subdomains: %d" % n_sub, "Test with f(x,y)=x:" + gc.fmt(round(forms[0], 10)),
                "Test with f(x,y)=x+y:" + gc.fmt(round(forms[1], 10))]
    want = [l for l in gc.golden_lines("poisson_sanity_check_03.output") if l and not l.startswith("Test with 1:")]
    assert out == want


@pytest.mark.parametrize("n_refine,n_sub,basis,p,vname", [(2, 64, "dgq", 1, "minsip"), (1, 30, "dgp", 2, "poisson"), (1, 25, "dgq", 3, "adm")])
def test_t3_mesh_parity_with_the_oracle(n_refine, n_sub, basis, p, vname):
    """Agglomerates of the unstructured mesh, entry by entry against the oracle (product mirror on the input side, the oracle
    reading the same file with its own reader): non-axis-aligned faces, irregular polytopes, 2-D kernels."""
    from test_host_golden import _t3_pair, VARIANTS
    import polydeal_amd as pa

    grid, ah, fe, og, oah, ofe = _t3_pair(n_refine, n_sub, basis, p, p + 1)
    pv, ov = VARIANTS[vname]
    pvar = pv(fe) if vname in ("poisson", "dr") else pv()
    ovar = ov(ofe) if vname in ("poisson", "dr") else ov()
    for diag_first in (True, False):
        rp, ci, vals = pa.assemble_dg_matrix(fe, ah, pvar, diag_first=diag_first)
        orp, oci, ref = po.assemble_csr(oah, ovar, diag_first=diag_first)
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        assert_parity(vals, ref, orp, oci, fe.n_dofs_per_cell)


@pytest.mark.parametrize("basis,p", [("dgq", 3), ("dgp", 3), ("dgq", 2)])
@pytest.mark.parametrize("world", [2, 3])
def test_row_kernel_staircase_rank_local_descriptions(world, basis, p):
    """Irregular (grown) agglomerates of Cartesian cells from the PRODUCT mirror, partitioned into `world` row ranges, every rank
    with its own rank-local description (pdh_problem.local = 1: own polytopes + ghost neighbours): the MULTI row kernel must
    reproduce the rows of the global assembly (faces cut by the partition are seen from the owned side only), and the global
    assembly the oracle's."""
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, 6, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    ah.define_grown_agglomerates(6, seed=7)
    fe = pa.FE_DGQ(3, p) if basis == "dgq" else pa.FE_AggloDGP(3, p)
    ah.initialize_fe_values(p + 1, p + 1)
    ah.distribute_agglomerated_dofs(fe)
    var = pa.SipVariant.poisson_example(fe)
    n, nA, N = fe.n_dofs_per_cell, ah.n_agglomerates, ah.n_dofs
    gflat = ah.flatten(var, True, True)
    ctx = pa.Context(0)
    ctx.set_problem(gflat)
    assert ctx.algorithm_in_use() == "rows"
    assert ctx.rows_kernel_in_use() == "terms"
    gvals = ctx.assemble()
    ctx.close()
    # oracle on the same agglomerates
    og = po.subdivided_hyper_cube(3, 6, 0.0, 1.0)
    oah = po.AgglomerationHandler(og)
    for P in range(nA):
        cells = ah.get_agglomerate(P)
        oah.define_agglomerate([cells[-1]] + cells[:-1])
    ofe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    oah.initialize_fe_values(p + 1, p + 1)
    oah.distribute_agglomerated_dofs(ofe)
    orp, oci, ref = po.assemble_csr(oah, po.variant_poisson_example(ofe))
    ga = gflat.arrays()
    assert np.array_equal(ga["rowptr"], orp) and np.array_equal(ga["colind"], oci)
    assert_parity(gvals, ref, orp, oci, n)
    splits = [row_range(nA, n, r, world)[0] for r in range(world)] + [N]
    for r in range(world):
        r0, r1 = splits[r], splits[r + 1]
        loc = ah.flatten_local(var, r0, r1, True, False, row_splits=splits)
        ctx = pa.Context(0)
        ctx.set_problem(loc, r0, r1)
        assert ctx.algorithm_in_use() == "rows"
        v = ctx.assemble()
        ctx.close()
        want = gvals[orp[r0]:orp[r1]]
        assert v.shape == want.shape
        assert np.max(np.abs(v - want)) <= 1e-13 * np.max(np.abs(gvals))


@pytest.mark.parametrize("basis,p", [("dgp", 3), ("dgq", 3), ("dgq", 2), ("dgp", 2), ("dgq", 1), ("dgp", 1)])
@pytest.mark.parametrize("kind,cells,per,vname,diag_first,nq", [("block", 4, 2, "poisson", True, 0), ("grown", 6, 6, "dr", False, 0), ("grown", 8, 8, "adm", True, 0),
                                                                ("block", 6, 3, "minsip", True, 0), ("block", 4, 2, "test", False, 7)])
def test_cartesian_description_points_generated_on_device(kind, cells, per, vname, diag_first, nq, basis, p):
    """pdh_set_problem_cartesian: the PRODUCT mirror describes the agglomerates of Cartesian cells without their quadrature points
    (every group of points named by its cell and local face), the points are generated on the device - the gather the reference
    times (source/agglomeration_handler.cc:622-707, 1103-1243).  Values against the oracle per block (own mesh + agglomerates
    rebuilt there), against the points-based description to rounding, and on row ranges with rank-local descriptions."""
    import polydeal_amd as pa
    from polydeal_amd.partition import row_range

    nq = nq or p + 1
    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    if kind == "block":
        ah.define_block_agglomerates(per)
    else:
        ah.define_grown_agglomerates(per, seed=cells)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(3, p)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = {"poisson": pa.SipVariant.poisson_example(fe), "dr": pa.SipVariant.diffusion_reaction(fe), "adm": pa.SipVariant.assemble_dg_matrix(),
           "minsip": pa.SipVariant.minimal_sip_example(), "test": pa.SipVariant.minimal_sip_test()}[vname]
    n, nA = fe.n_dofs_per_cell, ah.n_agglomerates
    cf = ah.flatten_cartesian(var, diag_first, True)
    assert cf.cartesian and cf.c.vq_x is None and cf.c.fq_x is None
    ctx = pa.Context(0)
    ctx.set_problem(cf)
    assert ctx.algorithm_in_use() == "rows" and ctx.rows_kernel_in_use() == "terms"
    vals = ctx.assemble()
    ctx.close()
    # the same problem described with its points
    pf = ah.flatten(var, diag_first, True)
    ca, pa_ = cf.arrays(), pf.arrays()
    for key in ("bbox", "dof_offset", "vq_ptr", "face_in", "face_out", "fq_ptr", "face_sigma", "rowptr", "colind"):
        assert np.array_equal(ca[key], pa_[key]), key
    ctx = pa.Context(0)
    ctx.set_problem(pf)
    vpts = ctx.assemble()
    ctx.close()
    assert np.max(np.abs(vals - vpts)) <= 1e-13 * np.max(np.abs(vpts))
    # the oracle on the same agglomerates
    og = po.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    oah = po.AgglomerationHandler(og)
    for P in range(nA):
        c_ = ah.get_agglomerate(P)
        oah.define_agglomerate([c_[-1]] + c_[:-1])
    ofe = po.FE_DGQ(3, p) if basis == "dgq" else po.FE_AggloDGP(3, p)
    oah.initialize_fe_values(nq, nq)
    oah.distribute_agglomerated_dofs(ofe)
    orp, oci, ref = po.assemble_csr(oah, variant(vname, ofe), diag_first=diag_first)
    assert np.array_equal(ca["rowptr"], orp) and np.array_equal(ca["colind"], oci)
    assert_parity(vals, ref, orp, oci, n)
    # rank-local compact descriptions of two row ranges reproduce the rows
    splits = [row_range(nA, n, r, 2)[0] for r in range(2)] + [ah.n_dofs]
    for r in range(2):
        loc = ah.flatten_cartesian(var, diag_first, False, splits[r], splits[r + 1], splits)
        ctx = pa.Context(0)
        ctx.set_problem(loc, splits[r], splits[r + 1])
        v = ctx.assemble()
        ctx.close()
        want = vals[orp[splits[r]]:orp[splits[r + 1]]]
        assert v.shape == want.shape and np.max(np.abs(v - want)) <= 1e-13 * np.max(np.abs(vals))


def test_cartesian_description_refuses_what_it_cannot_describe():
    """A distorted grid has no boxes (the mirror refuses to write the compact description); polytopes whose tables exceed the term
    kernels' LDS budget are refused by the library with PDH_EUNSUPPORTED and a message that says what to do."""
    import polydeal_amd as pa

    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, 4, 0.0, 1.0)
    grid.distort(0.1, 1)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(2)
    fe = pa.FE_AggloDGP(3, 2)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    with pytest.raises(Exception, match="box"):
        ah.flatten_cartesian(pa.SipVariant.poisson_example(fe))
    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, 8, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(4)  # 64 cells, up to 96 sub-faces per polytope
    fe = pa.FE_DGQ(3, 3)
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    cf = ah.flatten_cartesian(pa.SipVariant.poisson_example(fe))
    ctx = pa.Context(0)
    os.environ["PDH_TERMS_MERGE"] = "0"  # (summed over as given; merged into 2^3 cells and 2 x 2 sub-faces per plane they fit, below)
    try:
        with pytest.raises(pa.PdhError, match="points"):
            ctx.set_problem(cf)
    finally:
        del os.environ["PDH_TERMS_MERGE"]
    os.environ["PDH_TERMS_DGQ3"] = "0"
    try:
        ctx.set_problem(ah.flatten(pa.SipVariant.poisson_example(fe)))  # the context is still usable; pdh_rows.h takes the points
    finally:
        del os.environ["PDH_TERMS_DGQ3"]
    assert ctx.algorithm_in_use() == "rows" and ctx.rows_kernel_in_use() == "pieces"
    ref = ctx.assemble()
    ctx.set_problem(cf)
    assert ctx.rows_kernel_in_use() == "terms"
    st = ctx.terms_merge_stats()
    assert st["cells"] == 512 and st["cells_merged"] == 64 and st["sub_faces_merged"] * 4 == st["sub_faces"]
    got = ctx.assemble()
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))
    ctx.close()
