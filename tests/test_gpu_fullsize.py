"""Parity at BASELINE.json's FULL sizes through size-independent identities (the oracle cannot assemble these
sizes in seconds): for a globally continuous v the jump terms vanish, so with Nitsche boundary terms on the
unit cube [0,1]^d and uniform penalty sigma
    v = 1   : A v = 0 on rows of interior polytopes,  v^T A v = sigma * |dOmega| = 2 d sigma
    v = x_0 : v^T A v = int |grad x|^2 - 2 int_dO x d_n x + sigma int_dO x^2 = -1 + sigma (1 + (2d-2)/3)
These touch every kernel output (diagonal blocks, both coupling blocks of every face, CSR placement incl. the
deal.II diagonal-first layout).  The same identities are what test/polydeal/poisson_sanity_check_01..03 print
(there with boundary terms dropped)."""
import numpy as np
import pytest
import scipy.sparse as sp

import polydeal_amd as pa

pytestmark = pytest.mark.gpu


def coefficient_vectors(ah, fe):
    """Coefficients of v=1 and v=x_0 in every polytope's basis on its bounding box."""
    n = fe.n_dofs_per_cell
    nA = ah.n_agglomerates
    one = np.zeros((nA, n))
    x0 = np.zeros((nA, n))
    lo = np.zeros(nA)
    hi = np.zeros(nA)
    off = np.zeros(nA, dtype=np.int64)
    for P in range(nA):
        l, h = ah.bbox(P)
        lo[P], hi[P] = l[0], h[0]
        off[P] = ah.dof_indices(P)[0]
    if fe.basis == pa.PDH_BASIS_DGQ:
        p = fe.degree
        from oracle.polydeal_oracle import gauss_lobatto_nodes  # node positions only (test infrastructure)
        nodes = gauss_lobatto_nodes(p)
        i0 = np.arange(n) % (p + 1)  # x index, lexicographic x fastest
        one[:] = 1.0
        x0[:] = lo[:, None] + nodes[i0][None, :] * (hi - lo)[:, None]
    else:
        # Legendre: phi_0 = 1; the function with multi-index e_0 is index 1: sqrt(3)(2 xhat - 1)
        one[:, 0] = 1.0
        x0[:, 0] = 0.5 * (lo + hi)
        x0[:, 1] = (hi - lo) / (2.0 * np.sqrt(3.0))
    v1 = np.zeros(ah.n_dofs)
    vx = np.zeros(ah.n_dofs)
    idx = off[:, None] + np.arange(n)[None, :]
    v1[idx] = one
    vx[idx] = x0
    return v1, vx, off


def run_identities(dim, cells, block, fe, nq, unstructured_rules=None, distort=0.0, expect_alg=None):
    """unstructured_rules = name of the algorithm AUTO must then take: the quadrature points are declared unstructured
    (pdh_problem::vq_tensor_n = fq_tensor_n = -1), which keeps the kinds of the row kernel that need tensor rules out.
    distort > 0: interior vertices moved by up to distort * h - general hexahedra / quadrilaterals, polytopes of different
    diameters (sigma differs from face to face: the closed forms are then evaluated on the face quadrature itself)."""
    lg = cells.bit_length() - 1
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, lg)
    if distort:
        grid.distort(distort, 3)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(block)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = pa.SipVariant.poisson_example(fe)
    bdry_sigma_area = bdry_sigma_x2 = None
    if distort:
        flat = ah.flatten(var, True, True)
        arr = flat.arrays()
        bd = arr["face_out"] < 0
        w, x0q = arr["fq_w"], arr["fq_x"][0] if arr["fq_x"].ndim == 2 else arr["fq_x"][:len(arr["fq_w"])]
        fp = arr["fq_ptr"]
        # (per-face sums by reduceat: differences of a running sum over 14 M weights would carry 1e-11 of rounding themselves)
        area = np.add.reduceat(w, fp[:-1])
        x2 = np.add.reduceat(w * x0q * x0q, fp[:-1])
        bdry_sigma_area = float(np.sum(arr["face_sigma"][bd] * area[bd]))
        bdry_sigma_x2 = float(np.sum(arr["face_sigma"][bd] * x2[bd]))
        ctx = pa.Context(0)
        ctx.set_problem(flat)
        if expect_alg is not None:
            assert ctx.algorithm_in_use() == expect_alg, ctx.algorithm_in_use()
        vals = ctx.assemble()
        ctx.close()
        rp, ci = arr["rowptr"].copy(), arr["colind"].copy()
        del flat, arr
    elif unstructured_rules is None:
        rp, ci, vals = pa.assemble_dg_matrix(fe, ah, var, diag_first=True)
    else:
        flat = ah.flatten(var, True, True)
        flat.c.vq_tensor_n = flat.c.fq_tensor_n = -1
        ctx = pa.Context(0)
        ctx.set_problem(flat)
        assert ctx.algorithm_in_use() == unstructured_rules
        vals = ctx.assemble()
        ctx.close()
        arr = flat.arrays()
        rp, ci = arr["rowptr"].copy(), arr["colind"].copy()
        del flat
    assert np.all(np.isfinite(vals))
    A = sp.csr_matrix((vals, ci, rp), shape=(ah.n_dofs, ah.n_dofs))
    sigma = var.penalty_constant / ah.diameter(0)
    v1, vx, off = coefficient_vectors(ah, fe)
    y1 = A @ v1
    scale = float(np.max(np.abs(vals)))
    # interior polytopes: all bbox faces strictly inside the domain
    nb = cells // block
    n = fe.n_dofs_per_cell
    interior = np.ones(ah.n_agglomerates, dtype=bool)
    for P in range(ah.n_agglomerates):
        l, h = ah.bbox(P)
        if np.min(l) < 1e-12 or np.max(h) > 1 - 1e-12:
            interior[P] = False
    rows = (off[interior][:, None] + np.arange(n)[None, :]).ravel()
    assert interior.sum() == (nb - 2) ** dim
    assert np.max(np.abs(y1[rows])) <= 1e-12 * scale * n
    q1 = float(v1 @ y1)
    qx = float(vx @ (A @ vx))
    if distort:  # v = 1: sum over the boundary faces of sigma_F |F|; v = x_0: 1 - 2 + sum_F sigma_F int_F x_0^2
        # (bound: entries good to 1e-12 relative, coefficients |v| <= 1 - the quadratic forms are sums of 9e8 entries that cancel
        # to 1e-2 of their absolute sum, so the result itself carries fewer digits than an entry)
        tol = 1e-12 * float(np.sum(np.abs(vals)))
        assert abs(q1 - bdry_sigma_area) <= tol, (q1, bdry_sigma_area, tol)
        assert abs(qx - (-1.0 + bdry_sigma_x2)) <= tol, (qx, -1.0 + bdry_sigma_x2, tol)
        sample = np.unique(np.linspace(0, ah.n_dofs - 1, 400).astype(np.int64))
        S = A[sample][:, sample]
        assert abs(S - S.T).max() <= 1e-12 * scale
        return ah.n_dofs, len(vals)
    assert abs(q1 - 2 * dim * sigma) <= 1e-11 * 2 * dim * sigma
    exact_x = -1.0 + sigma * (1.0 + (2 * dim - 2) / 3.0)
    assert abs(qx - exact_x) <= 1e-11 * abs(exact_x)
    # symmetry on a sample of rows (full transpose is too heavy at this size)
    sample = np.unique(np.linspace(0, ah.n_dofs - 1, 400).astype(np.int64))
    S = A[sample][:, sample]
    assert abs(S - S.T).max() <= 1e-12 * scale
    return ah.n_dofs, len(vals)


def test_config2_2d_p2_4096_polytopes():
    """BASELINE.json configs[1]: 2-D unit square, 128^2 cells, 4096 polytopes of 2x2, p=2 (both bases)."""
    assert run_identities(2, 128, 2, pa.FE_AggloDGP(2, 2), 3)[0] == 24576
    assert run_identities(2, 128, 2, pa.FE_DGQ(2, 2), 3)[0] == 36864


def test_config3_3d_p3_32768_polytopes_dgp():
    """BASELINE.json configs[2] with FE_AggloDGP(3) (what examples/poisson.cc instantiates): 655 360 dofs."""
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_AggloDGP(3, 3), 4)
    assert n_dofs == 655360 and nnz == 89292800


def test_config3_3d_p3_32768_polytopes_dgq():
    """BASELINE.json configs[2] with FE_DGQ(3), (p+1)^3 = 64 dofs per polytope: the bench workload itself."""
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_DGQ(3, 3), 4)
    assert n_dofs == 2097152 and nnz == 914358272


def test_3d_p2_32768_polytopes_dgq_mixed_algorithm():
    """FE_DGQ(2), n = 27 (the element of BASELINE.json configs[3]) at 32 768 polytopes.  On this Cartesian mesh AUTO takes the
    row kernel (streamed kind); with the rules declared unstructured it takes the moment form for the diagonal blocks and the
    direct form for the coupling blocks - the identities see all three."""
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_DGQ(3, 2), 3)
    assert n_dofs == 884736 and nnz == 27 * 27 * 223232
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_DGQ(3, 2), 3, unstructured_rules="mixed")
    assert n_dofs == 884736 and nnz == 27 * 27 * 223232


def test_config3_forced_moment_form_fullsize():
    """The generic moment kernels (k_mdiag / k_moffdiag) at the bench size: AUTO now takes the row kernel on this Cartesian
    mesh, so the moment form is forced here and must satisfy the same identities (and agree with the row kernel)."""
    fe = pa.FE_DGQ(3, 3)
    grid = pa.BackgroundGrid.hyper_cube_refined(3, 0.0, 1.0, 5)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(2)
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    var = pa.SipVariant.poisson_example(fe)
    flat = ah.flatten(var, True, False)
    out = {}
    for alg in ("moment", "rows"):
        ctx = pa.Context(0)
        ctx.set_algorithm(alg)
        ctx.set_problem(flat)
        assert ctx.algorithm_in_use() == alg
        out[alg] = ctx.assemble()
        ctx.close()
    sc = np.max(np.abs(out["moment"]))
    assert np.max(np.abs(out["moment"] - out["rows"])) <= 1e-13 * sc


def test_config4_diffusion_reaction_128cubed_dgq2_one_gpu():
    """BASELINE.json configs[3] at its own size on ONE GPU: 128^3 cells, 262 144 polytopes of 2^3 cells, FE_DGQ(2) (n = 27),
    examples/diffusion_reaction.cc variant (sigma = 10 p^2 / h, owner id() < id(), reaction c = 0.5): 7 077 888 dofs,
    1.32 G non-zeros (10.6 GB of values).  Identities with the reaction term, for globally continuous v (jumps vanish):
        1^T A 1 = sigma |dOmega| + c |Omega| = 6 sigma + c
        x^T A x = -1 + sigma (1 + 4/3) + c / 3
        sum of (A 1) over the rows of an interior polytope = c |P|      (sum_i phi_i = 1)
    and the 8 Morton-octant row ranges (what 8 ranks of the reference own) reproduce the rows of the full assembly."""
    from polydeal_amd.partition import row_range

    fe = pa.FE_DGQ(3, 2)
    cells, block, dim = 128, 2, 3
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, 7)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(block)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    n = fe.n_dofs_per_cell
    assert ah.n_agglomerates == 262144 and ah.n_dofs == 7077888
    var = pa.SipVariant.diffusion_reaction(fe)
    c = var.reaction_c
    flat = ah.flatten(var, True, False)
    assert flat.nnz == n * n * (262144 + 2 * 3 * 64 * 64 * 63)
    ctx = pa.Context(0)
    ctx.set_problem(flat)
    vals = ctx.assemble()
    assert np.all(np.isfinite(vals))
    rp = flat.arrays()["rowptr"]
    # octants: contiguous eighths of the rows (masters are numbered in Morton order)
    worst = 0.0
    scale = float(np.max(np.abs(vals)))
    for r in range(8):
        r0, r1 = row_range(ah.n_agglomerates, n, r, 8)
        ctx.set_problem(flat, r0, r1)
        part = ctx.assemble()
        worst = max(worst, float(np.max(np.abs(part - vals[rp[r0]:rp[r1]]))))
        del part
    ctx.close()
    assert worst <= 1e-13 * scale, worst / scale
    _, ci = ah.sparsity_pattern(True)
    A = sp.csr_matrix((vals, ci, rp), shape=(ah.n_dofs, ah.n_dofs))
    sigma = var.penalty_constant / ah.diameter(0)
    v1, vx, off = coefficient_vectors(ah, fe)
    y1 = A @ v1
    q1 = float(v1 @ y1)
    qx = float(vx @ (A @ vx))
    e1 = 6 * sigma + c
    ex = -1.0 + sigma * (1.0 + 4.0 / 3.0) + c / 3.0
    assert abs(q1 - e1) <= 1e-11 * abs(e1), (q1, e1)
    assert abs(qx - ex) <= 1e-11 * abs(ex), (qx, ex)
    # interior polytopes: row sums of A 1 over the polytope = c |P|
    nb = cells // block
    ids = np.arange(ah.n_agglomerates)
    sums = y1.reshape(ah.n_agglomerates, n).sum(axis=1)  # rows of a polytope are contiguous, polytopes in dof order
    volP = (block / cells) ** 3
    # a polytope is interior iff its row sum needs no boundary term; count them through the value itself
    interior = np.abs(sums - c * volP) <= 1e-11 * scale
    assert interior.sum() == (nb - 2) ** 3, interior.sum()


def test_multi_row_kernel_beyond_the_resident_waves(monkeypatch):
    """MULTI row kernel (irregular agglomerates, staircase faces) on MORE polytopes than the device holds resident waves
    (32^3 cells in 4096 grown agglomerates; 2048 waves on 256 CUs): every wave works through several polytopes and re-uses its
    row of the coupling-moment scratch (PdhRows::m2c_scratch) - the small parity cases against the oracle give every wave
    one polytope.  Checked against the moment form (no scratch, no persistent waves; itself parity-tested against the oracle
    on the small cases) entry by entry, and through the size-independent identities  A 1 = 0 on interior rows,
    1^T A 1 = sigma |dOmega| with the polytope-wise penalty the caller variant sets.  (PDH_TERMS_DGQ3=0: AUTO would take the
    workgroup term kernel, which has no persistent waves.)"""
    monkeypatch.setenv("PDH_TERMS_DGQ3", "0")
    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, 32, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    ah.define_grown_agglomerates(8, seed=3)
    fe = pa.FE_DGQ(3, 3)
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    assert ah.n_agglomerates > 2048 + 1024
    var = pa.SipVariant.poisson_example(fe)
    flat = ah.flatten(var, True, True)
    out = {}
    for alg in ("rows", "moment"):
        ctx = pa.Context(0)
        ctx.set_algorithm(alg)
        ctx.set_problem(flat)
        assert ctx.algorithm_in_use() == alg
        ctx.poison_values()
        ctx.assemble_device()
        ctx.assemble_device()  # (a second pass over the same scratch rows)
        out[alg] = ctx.assemble()
        ctx.close()
    assert np.all(np.isfinite(out["rows"]))
    scale = float(np.max(np.abs(out["moment"])))
    assert float(np.max(np.abs(out["rows"] - out["moment"]))) <= 1e-12 * scale
    arr = flat.arrays()
    A = sp.csr_matrix((out["rows"], arr["colind"].copy(), arr["rowptr"].copy()), shape=(ah.n_dofs, ah.n_dofs))
    v1 = np.ones(ah.n_dofs)  # FE_DGQ: nodal basis, the constant has all coefficients one
    y1 = A @ v1
    n = fe.n_dofs_per_cell
    for P in range(0, ah.n_agglomerates, 7):
        l, h = ah.bbox(P)
        if np.min(l) > 1e-12 and np.max(h) < 1 - 1e-12:
            d0 = ah.dof_indices(P)[0]
            assert np.max(np.abs(y1[d0:d0 + n])) <= 1e-12 * scale * n
    sample = np.unique(np.linspace(0, ah.n_dofs - 1, 400).astype(np.int64))
    S = A[sample][:, sample]
    assert abs(S - S.T).max() <= 1e-12 * scale


def test_bench_strong_scaling_rehearsal_on_one_gpu():
    """`bench.py --gpus 4 --scaling strong` (the default scaling, BASELINE.json north_star: strong scaling to 8 GPUs) rehearsed on
    this box's ONE GPU: four ranks (gloo, all on device 0 - the box admits at most six processes on the card, so the 8-rank
    case is left to the driver's 8-GPU node) split the rows of one problem, each from its rank-local description; the JSON line
    must carry the strong-scaling fields, the global number of non-zeros, and a valid matrix on every rank's device values
    (rank 0's checksum against its closed form).  The timing is meaningless here."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--rehearse-on-one-gpu", "--cells", "16", "--steps", "2",
           "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--no-exchange-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["scaling"] == "strong" and d["n_gpus"] == 4
    n_agg, n = 8 ** 3, 64
    assert d["config"]["workload"].endswith("%d dofs, %d nnz" % (n_agg * n, n * n * (n_agg + 2 * 3 * 8 * 8 * 7)))
    assert "split in 4 contiguous ranges" in d["config"]["parallelism"]
    assert d["checksum"]["non_finite"] == 0 and d["checksum"]["rel_err"] < 1e-9
    assert d["value"] > 0 and d["roofline"]["kernel"] == "k_terms_wg"  # (rank-local descriptions: the workgroup term kernel)


def test_config3_distorted_mesh_fullsize():
    """The headline mesh with every interior vertex moved by up to 0.1 h (general hexahedra, non-planar faces: what the reference's
    own defaults look like - examples/3D_piston.cc:396-400, test/polydeal/exact_solutions_dgp.cc:306): AUTO leaves the row kernels
    (moment form for FE_DGQ(3), direct form for FE_AggloDGP(3)); the identities hold on any mesh and see every block."""
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_DGQ(3, 3), 4, distort=0.1, expect_alg="moment")
    assert n_dofs == 2097152 and nnz == 914358272
    n_dofs, nnz = run_identities(3, 64, 2, pa.FE_AggloDGP(3, 3), 4, distort=0.1, expect_alg="direct")
    assert n_dofs == 655360 and nnz == 89292800
