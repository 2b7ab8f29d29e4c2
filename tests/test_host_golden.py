"""Pins the PRODUCT's C++ host mirror (polydeal_amd/csrc/host/polydeal_host.h, via polydeal_amd.handler)
to the same golden outputs of the reference's tests as the oracle, and to the oracle itself."""
import numpy as np
import pytest

import golden_cases as gc
import polydeal_amd as pa
from flatten_oracle import flatten as oracle_flatten
from oracle import polydeal_oracle as po


def make(dim, refine, groups, singletons=True, fe=None, nq=1, nqf=1, lo=-1.0, hi=1.0):
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, lo, hi, refine)
    ah = pa.AgglomerationHandler(grid)
    if singletons:
        gc.define_with_singletons(ah, grid.n_cells, groups)
    else:
        for g in groups:
            ah.define_agglomerate(sorted(g))
    ah.initialize_fe_values(nq, nqf)
    ah.distribute_agglomerated_dofs(fe or pa.FE_DGQ(dim, 1))
    return grid, ah


def test_reinit_cell_face_quad_pts():
    _, ah = make(2, 3, gc.GROUPS_QUAD_PTS)
    assert gc.render_reinit_cell_face_quad_pts(ah) == gc.golden_lines("reinit_cell_face_quad_pts.output")[:-1]


def test_continuous_face_01_02():
    _, a0 = make(2, 2, gc.GROUPS_HALVES, singletons=False)
    _, a1 = make(2, 2, gc.GROUPS_2X2, singletons=False)
    out = gc.render_continuous_face_block(a0, None) + gc.render_continuous_face_block(a1, None)
    assert out == gc.golden_lines("continuous_face_01.output")[:-1]
    out = []
    for groups, label in zip(gc.GROUPS_CF02, ["End Test0", "End Test 1", "End Test 2"]):
        _, ah = make(2, 2, groups, singletons=False)
        out += gc.render_continuous_face_block(ah, label)
    assert out == gc.golden_lines("continuous_face_02.output")[: len(out)]


def test_master_master_and_neighbors():
    _, ah = make(2, 2, gc.GROUPS_MASTER_MASTER, singletons=False)
    assert gc.render_master_master(ah) == gc.golden_lines("reinit_cell_face_master_master.output")[:-1]
    _, ah = make(2, 2, gc.GROUPS_2X2, singletons=False)
    assert gc.render_neighbors_02(ah) == gc.golden_lines("agglomerated_neighbors_02.output")[:-1]
    assert gc.render_neighbors_03(ah) == gc.golden_lines("agglomerated_neighbors_03.output")[:-1]


def test_master_and_slaves():
    grid = pa.BackgroundGrid.hyper_cube_refined(2, -1.0, 1.0, 2)
    ah = pa.AgglomerationHandler(grid)
    for c in range(grid.n_cells):
        ah.define_agglomerate([c])
    ah.define_agglomerate([3, 6, 9, 12, 13])
    out = ["Cell with index: %d has associated value: %d" % (c, ah.master_slave_value(c)) for c in range(16)]
    assert out == gc.golden_lines("aggl_handler_master_and_slaves_01.output")[:-1]


def test_bbox_volume_perimeter():
    out = []
    for dim, cells in ((2, [3, 6, 9, 12, 13]), (3, [30, 58])):
        grid = pa.BackgroundGrid.hyper_cube_refined(dim, -1.0, 1.0, 2)
        ah = pa.AgglomerationHandler(grid)
        P = ah.define_agglomerate(cells)
        lo, hi = ah.bbox(P)
        out.append("p0: =" + " ".join(gc.fmt(x) for x in lo))
        out.append("p1: =" + " ".join(gc.fmt(x) for x in hi))
    assert out == gc.golden_lines("agg_handler_bbox_test.output")[:-1]
    _, a2 = make(2, 3, gc.GROUPS_FOUR)
    out = ["Sum is: " + gc.fmt(a2.volume_jxw_sum(P)) for P in range(4)]
    _, a3 = make(3, 3, [[463, 459]])
    out.append("Sum is: " + gc.fmt(a3.volume_jxw_sum(0)))
    assert out == gc.golden_lines("fe_space_on_bbox.output")[:-1]
    out = []
    for P in range(4):
        per = sum(a2.face_jxw_sum(P, f) for f in range(a2.n_faces_of(P)))
        out.append("Perimeter of polytope with index: %d is %s" % (P, gc.fmt(per)))
    assert out == gc.golden_lines("reinit_cell_face_01.output")[:-1]


def test_sparsity_dofs():
    _, ah = make(2, 3, gc.GROUPS_FOUR)
    assert gc.render_sparsity(ah) == gc.golden_lines("sparsity_agglomerated_tria.output")[:-1]
    grid, ah = make(2, 2, gc.GROUPS_2X2, singletons=False)
    assert gc.render_hp_structure(ah, grid.cell_vertices) == gc.golden_lines("hp_structure_01.output")[:-1]
    _, ah = make(2, 6, gc.GROUPS_POLY_ITER)
    out = ["dim = 2"] + gc.render_polytope_iterator_forward(ah, 7)
    assert out == gc.golden_lines("polytope_iterator.output")[: len(out)]
    rp, ci = ah.sparsity_pattern(diag_first=True)
    assert np.all(ci[rp[:-1]] == np.arange(ah.n_dofs))  # deal.II layout: diagonal first
    for r in (0, 17, ah.n_dofs - 1):
        row = ci[rp[r] + 1:rp[r + 1]]
        assert np.all(np.diff(row) > 0)


VARIANTS = {
    "adm": (pa.SipVariant.assemble_dg_matrix, po.variant_assemble_dg_matrix),
    "poisson": (pa.SipVariant.poisson_example, po.variant_poisson_example),
    "test": (pa.SipVariant.minimal_sip_test, po.variant_minimal_sip_test),
    "minsip": (pa.SipVariant.minimal_sip_example, po.variant_minimal_sip_example),
    "dr": (pa.SipVariant.diffusion_reaction, po.variant_diffusion_reaction),
}


@pytest.mark.parametrize("dim,refine,b,basis,p,nq,vname,distort", [
    (2, 3, 2, "dgq", 2, 3, "adm", 0.0),
    (2, 3, 2, "dgp", 2, 3, "poisson", 0.2),
    (3, 2, 2, "dgq", 1, 2, "test", 0.0),
    (3, 2, 2, "dgp", 3, 4, "poisson", 0.15),
    (2, 3, 4, "dgq", 1, 3, "minsip", 0.0),
    (3, 2, 2, "dgq", 2, 3, "dr", 0.0),
])
def test_flatten_matches_oracle(dim, refine, b, basis, p, nq, vname, distort):
    """Product flattening (C++) == oracle flattening (NumPy) on the same mesh: same tables, same
    quadrature data to round-off, same CSR pattern.  Un-distorted only for jittered grids the two
    sides use different RNGs, so there the oracle is fed the product's vertices."""
    grid = pa.BackgroundGrid.hyper_cube_refined(dim, 0.0, 1.0, refine)
    if distort:
        grid.distort(distort, seed=5)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(b)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(dim, p)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    pv, ov = VARIANTS[vname]
    pvar = pv(fe) if vname in ("poisson", "dr") else pv()
    flat = ah.flatten(pvar, diag_first=True, with_colind=True).arrays()

    og = po.hyper_cube_refined(dim, 0.0, 1.0, refine)
    for c in range(og.n_cells):
        og.vertices[c] = grid.cell_vertices(c)
    oah = po.AgglomerationHandler(og)
    for g in po.block_agglomerates(og, b):
        oah.define_agglomerate(g)
    ofe = (po.FE_DGQ if basis == "dgq" else po.FE_AggloDGP)(dim, p)
    oah.initialize_fe_values(nq, nq)
    oah.distribute_agglomerated_dofs(ofe)
    ovar = ov(ofe) if vname in ("poisson", "dr") else ov()
    ref = oracle_flatten(oah, ovar, diag_first=True)
    for k in ("dof_offset", "vq_ptr", "face_in", "face_out", "fq_ptr", "rowptr", "colind"):
        assert np.array_equal(np.asarray(ref[k]).ravel(), flat[k]), k
    for k in ("bbox", "vq_x", "vq_w", "fq_x", "fq_n", "fq_w", "fq_w_out", "face_sigma"):
        a, bb = np.asarray(ref[k], dtype=float).ravel(), flat[k]
        assert a.shape == bb.shape, k
        assert np.max(np.abs(a - bb)) <= 2e-15 * max(1.0, np.max(np.abs(a))), k


@pytest.mark.parametrize("dim,rep,lo,hi,b,basis,p,distort", [
    (3, (4, 2, 6), (0.0, 0.0, -1.0), (2.0, 1.0, 2.0), 2, "dgq", 2, 0.0),
    (2, (6, 3), (0.0, -1.0), (3.0, 1.0), 3, "dgp", 2, 0.1),
    (3, (2, 2, 4), (0.0, 0.0, 0.0), (1.0, 1.0, 2.0), 1, "dgp", 3, 0.0),
])
def test_flatten_on_subdivided_hyper_rectangle_matches_oracle(dim, rep, lo, hi, b, basis, p, distort):
    """GridGenerator::subdivided_hyper_rectangle grids (different cell counts and sizes per direction; the multi-GPU bench
    stacks the per-rank slabs along z): same tables from the product mirror and from the oracle."""
    grid = pa.BackgroundGrid.subdivided_hyper_rectangle(dim, rep, lo, hi)
    if distort:
        grid.distort(distort, seed=3)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(b)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(dim, p)
    ah.initialize_fe_values(p + 1, p + 1)
    ah.distribute_agglomerated_dofs(fe)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), diag_first=True, with_colind=True).arrays()
    og = po.subdivided_hyper_rectangle(dim, rep, lo, hi)
    assert og.n_cells == grid.n_cells
    for c in range(og.n_cells):
        if not distort:
            assert np.max(np.abs(og.vertices[c] - grid.cell_vertices(c))) <= 1e-15
        og.vertices[c] = grid.cell_vertices(c)
    oah = po.AgglomerationHandler(og)
    for g in po.block_agglomerates(og, b):
        oah.define_agglomerate(g)
    ofe = (po.FE_DGQ if basis == "dgq" else po.FE_AggloDGP)(dim, p)
    oah.initialize_fe_values(p + 1, p + 1)
    oah.distribute_agglomerated_dofs(ofe)
    ref = oracle_flatten(oah, po.variant_poisson_example(ofe), diag_first=True)
    for k in ("dof_offset", "vq_ptr", "face_in", "face_out", "fq_ptr", "rowptr", "colind"):
        assert np.array_equal(np.asarray(ref[k]).ravel(), flat[k]), k
    for k in ("bbox", "vq_x", "vq_w", "fq_x", "fq_n", "fq_w", "fq_w_out", "face_sigma"):
        a, bb = np.asarray(ref[k], dtype=float).ravel(), flat[k]
        assert a.shape == bb.shape, k
        assert np.max(np.abs(a - bb)) <= 4e-15 * max(1.0, np.max(np.abs(a))), k


def test_errors_are_reported():
    grid = pa.BackgroundGrid.hyper_cube_refined(2, 0.0, 1.0, 2)
    ah = pa.AgglomerationHandler(grid)
    ah.define_agglomerate([0, 1])
    with pytest.raises(pa.HostError):
        ah.distribute_agglomerated_dofs(pa.FE_DGQ(2, 1))  # not every cell agglomerated
    with pytest.raises(pa.HostError):
        pa.BackgroundGrid(2, 6, morton=True)  # Morton needs a power of two
    with pytest.raises(pa.HostError):
        ah.define_agglomerate([])


def test_reinit_cell_face_02_and_agglomerated_neighbors_01():
    """Two more face-enumeration goldens on irregular agglomerates of the 8x8 grid (540 + 928 lines)."""
    _, ah = make(2, 3, gc.GROUPS_RCF02)
    assert gc.render_reinit_cell_face_02(ah) == gc.golden_lines("reinit_cell_face_02.output")[:-1]
    _, ah = make(2, 3, gc.GROUPS_FOUR)
    assert gc.render_neighbors_02(ah) == gc.golden_lines("agglomerated_neighbors_01.output")[:-1]


def test_continuous_face_03():
    """test/polydeal/continuous_face_03.output on the product's host mirror."""
    grid = pa.BackgroundGrid.hyper_cube_refined(2, -1.0, 1.0, 3)
    ah = pa.AgglomerationHandler(grid)
    gc.define_continuous_face_03(ah, grid.n_cells)
    ah.initialize_fe_values(1, 1)
    ah.distribute_agglomerated_dofs(pa.FE_DGQ(2, 1))
    assert gc.render_continuous_face_03(ah) == gc.golden_lines("continuous_face_03.output")[:-1]


def test_rtree_level_counts_of_block_hierarchies():
    """rtree_mesh.output / extract_last_level.output pin polydeal_amd.levels.block_hierarchy (product side)."""
    from polydeal_amd.levels import block_hierarchy

    grid = pa.BackgroundGrid.hyper_cube_refined(2, 0.0, 1.0, 5)
    levels = block_hierarchy(grid, pa.FE_DGQ(2, 1), [32, 16, 8, 4, 2])
    sizes = [[ah.agglomerate_size(P) for P in range(ah.n_agglomerates)] for ah in levels]
    assert gc.render_rtree_levels(sizes[1:4], 1, "Extraction level = ", "Size of fine triangulation: ",
                                  "%d cells have subdomain id = %d") == gc.golden_lines("rtree_mesh.output")[:-1]
    assert gc.render_rtree_levels(sizes, 0, "Extract level: ", "Size of tria: ",
                                  "%d cells are composing agglomerate %d") == gc.golden_lines("extract_last_level.output")[:-1]


# ---------------------------------------------------------------------------------------------------------------------
# Unstructured background mesh: meshes/t3.msh of the reference (91 quadrilaterals, gmsh 4.1; tests/golden/t3.msh is that data
# file), read + refine_global as examples/minimal_SIP.cc:94-118 and test/polydeal/poisson_sanity_check_03.cc:103-113 do.
# ---------------------------------------------------------------------------------------------------------------------
T3 = __import__("os").path.join(gc.GOLDEN_DIR, "t3.msh")


def _t3_pair(n_refine, n_subdomains, basis="dgq", p=1, nq=3, seed=1):
    grid = pa.BackgroundGrid.read_msh(T3, n_refine)
    ah = pa.AgglomerationHandler(grid)
    ah.define_grown_agglomerates(max(1, grid.n_cells // n_subdomains), seed=seed)  # METIS stand-in (connected, irregular)
    fe = (pa.FE_DGQ if basis == "dgq" else pa.FE_AggloDGP)(2, p)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    og = po.read_msh(T3).refine_global(n_refine)
    oah = po.AgglomerationHandler(og)
    for P in range(ah.n_agglomerates):
        cells = ah.get_agglomerate(P)           # slaves in insertion order, then the master
        oah.define_agglomerate([cells[-1]] + cells[:-1])
    ofe = (po.FE_DGQ if basis == "dgq" else po.FE_AggloDGP)(2, p)
    oah.initialize_fe_values(nq, nq)
    oah.distribute_agglomerated_dofs(ofe)
    return grid, ah, fe, og, oah, ofe


def test_read_msh_and_refine_global_agree_with_the_oracle():
    grid = pa.BackgroundGrid.read_msh(T3, 0)
    og = po.read_msh(T3)
    assert grid.n_cells == og.n_cells == 91
    for k in (1, 2):
        grid = pa.BackgroundGrid.read_msh(T3, k)
        og.refine_global(1)
        assert grid.n_cells == og.n_cells == 91 * 4 ** k
        area = 0.0
        for c in range(og.n_cells):
            v = grid.cell_vertices(c)
            assert np.array_equal(v, og.vertices[c])
            assert all(grid.neighbor(c, f) == og.neighbor(c, f) for f in range(4))
            area += 0.5 * abs((v[3] - v[0])[0] * (v[2] - v[1])[1] - (v[3] - v[0])[1] * (v[2] - v[1])[0])
        assert abs(area - 1.0) < 1e-13  # the unit square
        # every interior edge is seen from both sides, each naming the other's face
        for c in range(og.n_cells):
            for f in range(4):
                nb = og.neighbor(c, f)
                if nb != po.INVALID:
                    assert og.neighbor(nb, og.neighbor_of_neighbor(f, c)) == c


@pytest.mark.parametrize("n_refine,n_sub,basis,p,vname", [(1, 20, "dgq", 1, "minsip"), (1, 40, "dgp", 2, "poisson"), (2, 50, "dgq", 2, "adm")])
def test_flatten_on_t3_mesh_matches_oracle(n_refine, n_sub, basis, p, vname):
    """Product flattening == oracle flattening on agglomerates of the unstructured mesh: both sides of every polytopal face see
    the same points in the same order (deal.II's face orientation; here: the direction of ascending global vertex numbers)."""
    grid, ah, fe, og, oah, ofe = _t3_pair(n_refine, n_sub, basis, p, p + 1)
    pv, ov = VARIANTS[vname]
    pvar = pv(fe) if vname in ("poisson", "dr") else pv()
    ovar = ov(ofe) if vname in ("poisson", "dr") else ov()
    flat = ah.flatten(pvar, diag_first=True, with_colind=True).arrays()
    ref = oracle_flatten(oah, ovar, diag_first=True)
    for k in ("dof_offset", "vq_ptr", "face_in", "face_out", "fq_ptr", "rowptr", "colind"):
        assert np.array_equal(np.asarray(ref[k]).ravel(), flat[k]), k
    for k in ("bbox", "vq_x", "vq_w", "fq_x", "fq_n", "fq_w", "fq_w_out", "face_sigma"):
        a, bb = np.asarray(ref[k], dtype=float).ravel(), flat[k]
        assert a.shape == bb.shape, k
        assert np.max(np.abs(a - bb)) <= 4e-15 * max(1.0, np.max(np.abs(a))), k
    # the two sides of a straight shared edge carry the same JxW point by point
    assert np.max(np.abs(flat["fq_w"] - flat["fq_w_out"])) <= 1e-15


_TWO_QUADS = """$MeshFormat
%s 0 8
$EndMeshFormat
$Nodes
1 8 1 8
2 1 0 8
1
2
3
4
5
6
7
8
0 0 0
1 0 0
1 1 0
0 1 0
3 0 0
4 0 0
4 1 0
3 1 0
$EndNodes
$Elements
1 2 1 2
2 1 3 2
1 1 2 3 4
2 5 6 7 8
$EndElements
"""


def test_read_msh_takes_format_4_1_only_and_grown_regions_need_a_seed_per_component(tmp_path):
    """a gmsh 4.0 file has another $Nodes / $Elements layout and must be refused, not misread; and a connected
    component of the mesh that received no seed must end the pocket-filling loop of the grown agglomerates with an error."""
    f41, f40 = tmp_path / "two_41.msh", tmp_path / "two_40.msh"
    f41.write_text(_TWO_QUADS % "4.1")
    f40.write_text(_TWO_QUADS % "4")
    with pytest.raises(Exception, match="4.1"):
        pa.BackgroundGrid.read_msh(str(f40), 0)
    grid = pa.BackgroundGrid.read_msh(str(f41), 0)
    assert grid.n_cells == 2
    ah = pa.AgglomerationHandler(grid)
    with pytest.raises(Exception, match="component"):
        ah.define_grown_agglomerates(2, seed=0)  # one region, two components
    ah2 = pa.AgglomerationHandler(pa.BackgroundGrid.read_msh(str(f41), 1))  # 8 cells, four per component
    ah2.define_grown_agglomerates(1, seed=0)  # a region per cell: every component has its seeds
    assert ah2.n_agglomerates == 8
