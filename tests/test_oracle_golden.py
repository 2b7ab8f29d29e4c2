"""Pins the oracle (oracle/polydeal_oracle.py) to the golden outputs of the reference's tests."""
import numpy as np
import pytest

import golden_cases as gc
from oracle import polydeal_oracle as po


def make(dim, refine, groups, singletons=True, fe=None, nq=1, nqf=1):
    grid = po.hyper_cube_refined(dim, -1.0, 1.0, refine)
    ah = po.AgglomerationHandler(grid)
    if singletons:
        gc.define_with_singletons(ah, grid.n_cells, groups)
    else:
        for g in groups:
            ah.define_agglomerate(sorted(g))
    ah.initialize_fe_values(nq, nqf)
    ah.distribute_agglomerated_dofs(fe or po.FE_DGQ(dim, 1))
    return grid, ah


def test_reinit_cell_face_quad_pts():
    _, ah = make(2, 3, gc.GROUPS_QUAD_PTS)
    assert gc.render_reinit_cell_face_quad_pts(ah) == gc.golden_lines("reinit_cell_face_quad_pts.output")[:-1]


def test_face_qpoints_match_both_sides():
    # the 1e-15 assert of reinit_cell_face_quad_pts.cc:129-138
    _, ah = make(2, 3, gc.GROUPS_QUAD_PTS, nq=2, nqf=2)
    for P in range(ah.n_agglomerates):
        for f in range(ah.n_faces_of(P)):
            if not ah.at_boundary(P, f):
                Q = ah.neighbor(P, f)
                f0, f1 = ah.reinit_interface(P, Q, f, ah.neighbor_of_agglomerated_neighbor(P, f))
                assert np.max(np.abs(f0["x"] - f1["x"])) < 1e-15
                assert np.max(np.abs(f0["normal"] + f1["normal"])) < 1e-15


def test_continuous_face_01():
    _, a0 = make(2, 2, gc.GROUPS_HALVES, singletons=False)
    _, a1 = make(2, 2, gc.GROUPS_2X2, singletons=False)
    out = gc.render_continuous_face_block(a0, None) + gc.render_continuous_face_block(a1, None)
    assert out == gc.golden_lines("continuous_face_01.output")[:-1]


def test_continuous_face_02_tests_0_to_2():
    out = []
    for groups, label in zip(gc.GROUPS_CF02, ["End Test0", "End Test 1", "End Test 2"]):
        _, ah = make(2, 2, groups, singletons=False)
        out += gc.render_continuous_face_block(ah, label)
    gold = gc.golden_lines("continuous_face_02.output")
    assert out == gold[: len(out)]
    assert len(out) == 366  # Test 3 (METIS partition, lines 367-952) is not reproducible offline


def test_master_master():
    _, ah = make(2, 2, gc.GROUPS_MASTER_MASTER, singletons=False)
    assert gc.render_master_master(ah) == gc.golden_lines("reinit_cell_face_master_master.output")[:-1]


def test_agglomerated_neighbors_02_03():
    _, ah = make(2, 2, gc.GROUPS_2X2, singletons=False)
    assert gc.render_neighbors_02(ah) == gc.golden_lines("agglomerated_neighbors_02.output")[:-1]
    assert gc.render_neighbors_03(ah) == gc.golden_lines("agglomerated_neighbors_03.output")[:-1]


def test_master_and_slaves():
    # aggl_handler_master_and_slaves_01.cc: singletons first, then {3,6,9,12,13} re-agglomerated
    grid = po.hyper_cube_refined(2, -1.0, 1.0, 2)
    ah = po.AgglomerationHandler(grid)
    for c in range(grid.n_cells):
        ah.define_agglomerate([c])
    ah.define_agglomerate([3, 6, 9, 12, 13])
    out = ["Cell with index: %d has associated value: %d" % (c, ah.master_slave_value(c)) for c in range(16)]
    assert out == gc.golden_lines("aggl_handler_master_and_slaves_01.output")[:-1]


def test_bbox():
    # agg_handler_bbox_test.cc: 2D refine 2 {3,6,9,12,13}; 3D refine 2 {30,58}
    out = []
    for dim, cells in ((2, [3, 6, 9, 12, 13]), (3, [30, 58])):
        grid = po.hyper_cube_refined(dim, -1.0, 1.0, 2)
        ah = po.AgglomerationHandler(grid)
        P = ah.define_agglomerate(cells)
        lo, hi = ah.bbox(P)
        out.append("p0: =" + " ".join(gc.fmt(x) for x in lo))
        out.append("p1: =" + " ".join(gc.fmt(x) for x in hi))
    assert out == gc.golden_lines("agg_handler_bbox_test.output")[:-1]


def test_fe_space_on_bbox_volume_sums():
    _, a2 = make(2, 3, gc.GROUPS_FOUR)
    out = ["Sum is: " + gc.fmt(a2.volume_jxw_sum(P)) for P in range(4)]
    _, a3 = make(3, 3, [[463, 459]])
    out.append("Sum is: " + gc.fmt(a3.volume_jxw_sum(0)))
    assert out == gc.golden_lines("fe_space_on_bbox.output")[:-1]


def test_reinit_cell_face_01_perimeters():
    _, ah = make(2, 3, gc.GROUPS_FOUR)
    out = []
    for P in range(4):
        per = sum(ah.face_jxw_sum(P, f) for f in range(ah.n_faces_of(P)))
        out.append("Perimeter of polytope with index: %d is %s" % (P, gc.fmt(per)))
    assert out == gc.golden_lines("reinit_cell_face_01.output")[:-1]


def test_sparsity_agglomerated_tria():
    _, ah = make(2, 3, gc.GROUPS_FOUR)
    assert gc.render_sparsity(ah) == gc.golden_lines("sparsity_agglomerated_tria.output")[:-1]


def test_hp_structure_01():
    grid, ah = make(2, 2, gc.GROUPS_2X2, singletons=False)
    out = gc.render_hp_structure(ah, lambda c: grid.vertices[c])
    assert out == gc.golden_lines("hp_structure_01.output")[:-1]


def test_polytope_iterator_dofs():
    _, ah = make(2, 6, gc.GROUPS_POLY_ITER)
    out = ["dim = 2"] + gc.render_polytope_iterator_forward(ah, 7)
    gold = gc.golden_lines("polytope_iterator.output")
    assert out == gold[: len(out)]


# ---- operator-level known answers ---------------------------------------------------------------
@pytest.mark.parametrize("dim", [2, 3])
def test_minimal_SIP_Poisson_identity(dim):
    """test/polydeal/minimal_SIP_Poisson.cc:486-509: agglomerated 2x2(x2) blocks == coarse standard
    SIP, entry by entry to 1e-13 (FE_DGQ(1), penalty 20, h_f = 1, QGauss(3))."""
    fe = po.FE_DGQ(dim, 1)
    var = po.variant_minimal_sip_test()
    gs = po.hyper_cube_refined(dim, -1.0, 1.0, 1 if dim == 2 else 0)
    std = po.AgglomerationHandler(gs)
    for c in range(gs.n_cells):
        std.define_agglomerate([c])
    std.initialize_fe_values(3, 3)
    std.distribute_agglomerated_dofs(fe)
    ga = po.hyper_cube_refined(dim, -1.0, 1.0, 2 if dim == 2 else 1)
    agg = po.AgglomerationHandler(ga)
    groups = gc.GROUPS_2X2 if dim == 2 else [list(range(8))]
    for g in groups:
        agg.define_agglomerate(g)
    agg.initialize_fe_values(3, 3)
    agg.distribute_agglomerated_dofs(fe)
    A_std = po.assemble_dense(std, var)
    A_agg = po.assemble_dense(agg, var)
    assert A_std.shape == A_agg.shape
    assert np.max(np.abs(A_std - A_agg)) < 1e-13
    # and through the CSR scatter with deal.II's diagonal-first layout
    rp, ci, va = po.assemble_csr(agg, var, diag_first=True)
    assert np.max(np.abs(po.csr_to_dense(rp, ci, va, agg.n_dofs) - A_agg)) < 1e-13
    assert np.all(ci[rp[:-1]] == np.arange(agg.n_dofs))


def _sanity_forms(ah, var):
    """v^T A v with boundary terms dropped, v in {x, x+y, 1} (poisson_sanity_check_01/03)."""
    A = po.assemble_dense(ah, var)
    res = []
    for f in (lambda x: x[:, 0], lambda x: x[:, 0] + x[:, 1], lambda x: np.ones(len(x))):
        v = po.interpolate_nodal(ah, f)
        res.append(float(v @ A @ v))
    return res


@pytest.mark.parametrize("groups", [gc.GROUPS_FOUR, gc.GROUPS_QUAD_PTS])
def test_poisson_sanity_identities(groups):
    """poisson_sanity_check_03.output: 1, 2, ~1e-14 for any agglomeration of the unit square."""
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 3)
    ah = po.AgglomerationHandler(grid)
    gc.define_with_singletons(ah, grid.n_cells, groups)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    var = po.SipVariant("sanity", 10.0, "index", "diameter_in", boundary="zero")
    x, xy, one = _sanity_forms(ah, var)
    assert abs(x - 1.0) < 1e-12 and abs(xy - 2.0) < 1e-12 and abs(one) < 1e-12


@pytest.mark.parametrize("fe_cls,p", [(po.FE_DGQ, 1), (po.FE_AggloDGP, 1), (po.FE_AggloDGP, 2), (po.FE_DGQ, 2)])
def test_exact_solution_reproduced_on_distorted_grid(fe_cls, p):
    """exact_solutions_dgp.cc:685-704: a solution inside the space is reproduced to ~1e-14
    (randomly distorted 4x4 grid, FE_AggloDGP p=1,2).  Solved densely; Nitsche RHS assembled here."""
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 2).distort(0.25, seed=1)
    ah = po.AgglomerationHandler(grid)
    for g in [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]:
        ah.define_agglomerate(g)
    fe = fe_cls(2, p)
    ah.initialize_fe_values(2 * p + 1, 2 * p + 1)
    ah.distribute_agglomerated_dofs(fe)
    var = po.variant_poisson_example(fe)
    A = po.assemble_dense(ah, var)
    u_ex = (lambda x: x[:, 0] + x[:, 1] - 1.0) if p == 1 else (lambda x: x[:, 0] ** 2 + x[:, 0] * x[:, 1])
    lap = 0.0 if p == 1 else -2.0  # -Laplace u
    b = np.zeros(ah.n_dofs)
    for P in range(ah.n_agglomerates):
        fv = ah.reinit(P)
        b[ah.dof_indices(P)] += np.einsum("qi,q->i", fv["val"], fv["JxW"]) * lap
        for f in range(ah.n_faces_of(P)):
            if ah.at_boundary(P, f):
                ff = ah.reinit_face(P, f)
                sig = po.face_sigma(ah, var, P)
                g = np.einsum("qic,qc->qi", ff["grad"], ff["normal"])
                ue = u_ex(ff["x"])
                b[ah.dof_indices(P)] += np.einsum("qi,q->i", sig * ff["val"] - g, ue * ff["JxW"])
    u = np.linalg.solve(A, b)
    err = 0.0
    for P in range(ah.n_agglomerates):
        fv = ah.reinit(P)
        err += np.sum((fv["val"] @ u[ah.dof_indices(P)] - u_ex(fv["x"])) ** 2 * fv["JxW"])
    assert np.sqrt(err) < 1e-12
    assert np.max(np.abs(A - A.T)) < 1e-11 * np.max(np.abs(A))


def _poisson_test_setup():
    """test/polydeal/poisson.cc:122-241: [-1,1]^2, refine_global(6), 7 two-cell agglomerates + singletons,
    FE_DGQ(1), QGauss(3), penalty 20, hf = edge length of the master cell (:321), index()<index()."""
    grid = po.hyper_cube_refined(2, -1.0, 1.0, 6)
    ah = po.AgglomerationHandler(grid)
    gc.define_with_singletons(ah, grid.n_cells, gc.GROUPS_POLY_ITER)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    hf = 2.0 / 64
    var = po.SipVariant("test/poisson.cc", 20.0 / hf, "index", "one")
    return grid, ah, var


def poisson_l2_error(grid, ah, rowptr, colind, values):
    """RHS (poisson.cc:286-303), direct solve (:492-497), interpolate_to_fine_grid + integrate_difference with
    QGauss(1) (:505-523): the number the reference prints as 'L2 error:0.00647702' (poisson.output)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    pi = np.pi
    b = np.zeros(ah.n_dofs)
    for P in range(ah.n_agglomerates):
        fv = ah.reinit(P)
        f = 8 * pi * pi * np.sin(2 * pi * fv["x"][:, 0]) * np.sin(2 * pi * fv["x"][:, 1])
        b[ah.dof_indices(P)] += np.einsum("qi,q->i", fv["val"], f * fv["JxW"])
    A = sp.csr_matrix((values, colind, rowptr), shape=(ah.n_dofs, ah.n_dofs))
    u = spla.spsolve(A.tocsc(), b)
    err2 = 0.0
    for P in range(ah.n_agglomerates):
        coef = u[ah.dof_indices(P)]
        for cell in ah.get_agglomerate(P):
            V = grid.vertices[cell]  # 4 vertices: nodal interpolation onto FE_DGQ(1) of the fine cell
            val, _ = ah.fe.shape(ah.real_to_unit(P, V))
            uh_mid = np.mean(val @ coef)  # bilinear interpolant at the midpoint (QGauss(1))
            mid = V.mean(axis=0)
            h2 = (V[1, 0] - V[0, 0]) * (V[2, 1] - V[0, 1])
            err2 += h2 * (uh_mid - np.sin(2 * pi * mid[0]) * np.sin(2 * pi * mid[1])) ** 2
    return np.sqrt(err2)


def test_poisson_output_L2_error():
    """End-to-end known answer produced by the reference itself: test/polydeal/poisson.output."""
    grid, ah, var = _poisson_test_setup()
    rp, ci, va = po.assemble_csr(ah, var, diag_first=False)
    err = poisson_l2_error(grid, ah, rp, ci, va)
    assert "L2 error:" + gc.fmt(err) == gc.golden_lines("poisson.output")[0]


def test_reinit_cell_face_02_and_agglomerated_neighbors_01():
    """Two more face-enumeration goldens on irregular agglomerates of the 8x8 grid (540 + 928 lines)."""
    _, ah = make(2, 3, gc.GROUPS_RCF02)
    assert gc.render_reinit_cell_face_02(ah) == gc.golden_lines("reinit_cell_face_02.output")[:-1]
    _, ah = make(2, 3, gc.GROUPS_FOUR)
    assert gc.render_neighbors_02(ah) == gc.golden_lines("agglomerated_neighbors_01.output")[:-1]


def test_compute_global_error_known_answers():
    """oracle.compute_global_error (include/poly_utils.h:1647-1750): zero for a function in the polytopal space,
    and the analytic value for u_h = 0 (||xy||_L2 = 1/3, |xy|_H1 = sqrt(2/3) on the unit square)."""
    fe = po.FE_DGQ(2, 2)
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 3)
    ah = po.AgglomerationHandler(grid)
    for g in po.block_agglomerates(grid, 2):
        ah.define_agglomerate(g)
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(fe)
    f = lambda x: x[:, 0] * x[:, 1]
    df = lambda x: np.stack([x[:, 1], x[:, 0]], axis=1)
    l2, h1 = po.compute_global_error(ah, po.interpolate_nodal(ah, f), f, df)
    assert l2 < 1e-14 and h1 < 1e-13
    l2, h1 = po.compute_global_error(ah, np.zeros(ah.n_dofs), f, df)
    assert abs(l2 - 1.0 / 3.0) < 1e-14 and abs(h1 - np.sqrt(2.0 / 3.0)) < 1e-14


def _nested_pair(dim, lg, b_coarse, b_fine, p, distort=0.0):
    fe = po.FE_DGQ(dim, p)
    grid = po.hyper_cube_refined(dim, 0.0, 1.0, lg)
    if distort:
        grid.distort(distort, seed=3)
    out = []
    for b in (b_coarse, b_fine):
        ah = po.AgglomerationHandler(grid)
        for g in po.block_agglomerates(grid, b):
            ah.define_agglomerate(g)
        ah.initialize_fe_values(p + 1, p + 1)
        ah.distribute_agglomerated_dofs(fe)
        out.append(ah)
    return out


@pytest.mark.parametrize("dim,lg,p,distort", [(2, 3, 1, 0.0), (2, 3, 3, 0.2), (3, 2, 2, 0.1)])
def test_injection_matrix_reproduces_the_coarse_function(dim, lg, p, distort):
    """The property test/polydeal/distributed_injection_01.cc checks (its .output prints 'Norm of error(L2):
    9.97775e-16' / 2.75792e-15): injecting the coarse interpolant of a function of the polytopal space gives the
    fine interpolant.  Here with a polynomial of degree p per direction on block hierarchies."""
    coarse, fine = _nested_pair(dim, lg, 4, 2, p, distort)
    f = lambda x: 1.0 + x[:, 0] ** p - 0.5 * x[:, 1] ** p * x[:, 0] + (x[:, -1] ** p if dim == 3 else 0.0)
    M = po.fill_injection_matrix(coarse, fine)
    err = M @ po.interpolate_nodal(coarse, f) - po.interpolate_nodal(fine, f)
    assert np.max(np.abs(err)) < 5e-14
    assert np.allclose(M.sum(axis=1), 1.0, atol=1e-13)  # partition of unity: rows sum to one


def test_continuous_face_03():
    """test/polydeal/continuous_face_03.output (834 lines): 8x8 grid, singletons first, then {36..39}, {18,24,25}, {3,6}."""
    grid = po.hyper_cube_refined(2, -1.0, 1.0, 3)
    ah = po.AgglomerationHandler(grid)
    gc.define_continuous_face_03(ah, grid.n_cells)
    ah.initialize_fe_values(1, 1)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    assert gc.render_continuous_face_03(ah) == gc.golden_lines("continuous_face_03.output")[:-1]


def test_poisson_sanity_check_02():
    """test/polydeal/poisson_sanity_check_02.output: two half-square polytopes, v^T A v = 2 (step function) and 1 (|x - 1/2|)."""
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 1)
    ah = po.AgglomerationHandler(grid)
    ah.define_agglomerate([0, 2])
    ah.define_agglomerate([1, 3])
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    A = po.assemble_dense(ah, po.variant_minimal_sip_example())  # poisson_sanity_check_02.cc:214-262: 10 max(1/h), zeroed boundary
    out = []
    for name, f in (("Step function", lambda x: (x[:, 0] >= 0.5) * 1.0), ("V function", lambda x: np.abs(x[:, 0] - 0.5))):
        v = po.interpolate_nodal(ah, f)
        out.append("Test with %s = %s" % (name, gc.fmt(float(v @ A @ v))))
    assert out == gc.golden_lines("poisson_sanity_check_02.output")[:-1]


def test_rtree_level_counts_of_block_hierarchies():
    """test/polydeal/rtree_mesh.output and extract_last_level.output: on the 32x32 hyper_cube the R-tree levels the reference
    extracts (include/agglomerator.h:389-434) are 4^l agglomerates of 1024 / 4^l cells - exactly the block hierarchy that
    stands in for them here (blocks of 32, 16, 8, 4, 2 cells per direction)."""
    grid = po.hyper_cube_refined(2, 0.0, 1.0, 5)
    sizes = [[len(g) for g in po.block_agglomerates(grid, b)] for b in (32, 16, 8, 4, 2)]
    assert gc.render_rtree_levels(sizes[1:4], 1, "Extraction level = ", "Size of fine triangulation: ",
                                  "%d cells have subdomain id = %d") == gc.golden_lines("rtree_mesh.output")[:-1]
    assert gc.render_rtree_levels(sizes, 0, "Extract level: ", "Size of tria: ",
                                  "%d cells are composing agglomerate %d") == gc.golden_lines("extract_last_level.output")[:-1]


@pytest.mark.parametrize("n_sub", [50, 300])
def test_poisson_sanity_check_03_on_t3_mesh(n_sub):
    """test/polydeal/poisson_sanity_check_03.cc on ITS mesh: meshes/t3.msh (tests/golden/t3.msh, 91 quadrilaterals), refined 3
    times, N agglomerates - the reference partitions with METIS, here regions grown over the cell graph; the printed identities
    (v^T A v = 1, 2 for v = x, x + y and ~1e-14 for v = 1, boundary terms dropped, FE_DGQ(1), QGauss(3): :166-172) hold for ANY
    agglomeration.  Oracle only (the GPU suite runs all six N of the reference's output file through the HIP path)."""
    import os
    grid = po.read_msh(os.path.join(gc.GOLDEN_DIR, "t3.msh")).refine_global(3)
    assert grid.n_cells == 5824
    rng = np.random.default_rng(n_sub)
    owner = -np.ones(grid.n_cells, dtype=np.int64)
    seeds = rng.choice(grid.n_cells, size=n_sub, replace=False)
    front = [[int(s)] for s in seeds]
    owner[seeds] = np.arange(n_sub)
    left = grid.n_cells - n_sub
    while left:
        for k in range(n_sub):
            while front[k]:
                c = front[k][0]
                free = [x for x in (grid.neighbor(c, f) for f in range(4)) if x != po.INVALID and owner[x] < 0]
                if not free:
                    front[k].pop(0)
                    continue
                owner[free[0]] = k
                front[k].append(int(free[0]))
                left -= 1
                break
    ah = po.AgglomerationHandler(grid)
    for k in range(n_sub):
        ah.define_agglomerate(sorted(np.nonzero(owner == k)[0].tolist()))
    ah.initialize_fe_values(3, 3)
    ah.distribute_agglomerated_dofs(po.FE_DGQ(2, 1))
    var = po.SipVariant("sanity", 10.0, "index", "diameter_in", boundary="zero")
    x, xy, one = _sanity_forms(ah, var)
    lines = gc.golden_lines("poisson_sanity_check_03.output")
    at = lines.index("N subdomains: %d" % n_sub)
    assert "Test with f(x,y)=x:" + gc.fmt(round(x, 10)) == lines[at + 1]
    assert "Test with f(x,y)=x+y:" + gc.fmt(round(xy, 10)) == lines[at + 2]
    assert abs(x - 1.0) < 1e-12 and abs(xy - 2.0) < 1e-12 and abs(one) < 1e-12
