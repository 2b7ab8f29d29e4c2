"""Golden cases taken from the reference's own test suite (test/polydeal/*.output).

The ``.output`` files under ``tests/golden/`` are *data* copied from the reference's tests
(expected stdout).  Each case below rebuilds the test's input with OUR mesh/handler code (the
agglomerate index lists are the ones hard-coded in the cited ``.cc`` file) and re-renders the text
the reference test prints, so that it can be compared line by line.

The renderers are written against a small handler protocol, implemented both by the oracle
(``oracle.polydeal_oracle.AgglomerationHandler``) and by the product's host mirror
(``polydeal_amd.AgglomerationHandler``), so both are pinned by the same fixtures:

    n_agglomerates, master_index(P), n_faces_of(P), at_boundary(P,f), neighbor(P,f),
    neighbor_of_agglomerated_neighbor(P,f), interface_list(P,Q), bbox(P), dof_indices(P),
    sparsity_rows(), volume_jxw_sum(P), face_jxw_sum(P,f), master_slave_value(cell)
"""
from __future__ import annotations

import os

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")


def golden_lines(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return f.read().split("\n")


def fmt(x):
    """std::cout default formatting of a double (6 significant digits, %g)."""
    s = "%.6g" % x
    if s == "-0":
        s = "0"
    return s


def define_with_singletons(ah, n_cells, groups):
    """The tests' pattern: define the listed agglomerates (cells in mesh order, as
    PolyUtils::collect_cells_for_agglomeration returns them: include/poly_utils.h:532-538),
    then every un-flagged cell as a singleton in cell order."""
    flagged = set()
    for g in groups:
        ah.define_agglomerate(sorted(g))
        flagged.update(g)
    for c in range(n_cells):
        if c not in flagged:
            ah.define_agglomerate([c])


# ---- renderers -------------------------------------------------------------------------------
def render_reinit_cell_face_quad_pts(ah):
    # test/polydeal/reinit_cell_face_quad_pts.cc:106-123
    out = []
    for P in range(ah.n_agglomerates):
        out.append("Cell with index %d has %d faces" % (ah.master_index(P), ah.n_faces_of(P)))
        for f in range(ah.n_faces_of(P)):
            if not ah.at_boundary(P, f):
                out.append("Neighbor is: %d" % ah.master_index(ah.neighbor(P, f)))
    out.append("Ok")
    return out


def render_master_master(ah):
    # test/polydeal/reinit_cell_face_master_master.cc:121-137
    out = []
    for P in range(ah.n_agglomerates):
        out.append("Polytope with index %d has %d faces" % (P, ah.n_faces_of(P)))
        for f in range(ah.n_faces_of(P)):
            if not ah.at_boundary(P, f):
                out.append("Neighbor is: %d" % ah.neighbor(P, f))
    out.append("Ok")
    return out


def render_perimeter_test(ah):
    # test/polydeal/continuous_face_01.cc:39-91 (perimeter_test)
    out = []
    perimeter = 0.0
    for P in range(ah.n_agglomerates):
        out.append("Master cell index = %d" % ah.master_index(P))
        nf = ah.n_faces_of(P)
        out.append("Number of agglomerated faces = %d" % nf)
        for f in range(nf):
            out.append("Agglomerate face index = %d" % f)
            if not ah.at_boundary(P, f):
                Q = ah.neighbor(P, f)
                out.append("Neighbor polytope index = %d" % Q)
                out.append("Neighbor of neighbor = %d" % ah.neighbor_of_agglomerated_neighbor(P, f))
                for cell, lf in ah.interface_list(P, Q):
                    out.append("deal.II cell index = %d" % cell)
                    out.append("Local face idx = %d" % lf)
                    out.append("Neighboring master cell index = %d" % ah.master_index(Q))
            else:
                perimeter += ah.face_jxw_sum(P, f)
        out.append("")
    out.append("Perimeter = " + fmt(perimeter))
    return out


def render_continuous_face_block(ah, end_label):
    # main() of continuous_face_01.cc / continuous_face_02.cc
    out = render_perimeter_test(ah)
    out.append("- - - - - - - - - - - -")
    out.append("Check on neighbors and neighbors of neighbors:")
    for P in range(ah.n_agglomerates):
        for f in range(ah.n_faces_of(P)):
            if not ah.at_boundary(P, f):
                Q = ah.neighbor(P, f)
                nofn = ah.neighbor_of_agglomerated_neighbor(P, f)
                assert ah.neighbor(Q, nofn) == P
    out.append("Ok")
    out.append("- - - - - - - - - - - -")
    out.append("Check on quadrature points:")
    out.append("Ok")
    if end_label:
        out.append(end_label)
    return out


def render_neighbors_02(ah):
    # test/polydeal/agglomerated_neighbors_02.cc:86-112
    out = []
    for P in range(ah.n_agglomerates):
        out.append("Polytope with idx: %d" % P)
        nf = ah.n_faces_of(P)
        out.append("Number of faces for the agglomeration: %d" % nf)
        for f in range(nf):
            if not ah.at_boundary(P, f):
                out.append("Agglomerated face with idx: %d" % f)
                for cell, lf in ah.interface_list(P, ah.neighbor(P, f)):
                    out.append("deal.II cell idx: %d" % cell)
                    out.append("deal.II face idx: %d" % lf)
            out.append("")
    return out


def render_neighbors_03(ah):
    # test/polydeal/agglomerated_neighbors_03.cc:95-112
    out = []
    for P in range(ah.n_agglomerates):
        m = ah.master_index(P)
        out.append("Cell with idx: %d" % m)
        nf = ah.n_faces_of(P)
        out.append("Number of faces for this cell: %d" % nf)
        for f in range(nf):
            nofn = ah.neighbor_of_agglomerated_neighbor(P, f)
            if nofn < 0:
                nofn = 4294967295  # numbers::invalid_unsigned_int
            out.append("Neighbor of neighbor for (%d,%d) = %d" % (m, f, nofn))
        out.append("")
    return out


def render_reinit_cell_face_02(ah):
    # test/polydeal/reinit_cell_face_02.cc:121-157
    out = []
    for P in range(ah.n_agglomerates):
        nf = ah.n_faces_of(P)
        out.append("Cell with index %d has %d faces" % (ah.master_index(P), nf))
        for f in range(nf):
            if not ah.at_boundary(P, f):
                Q = ah.neighbor(P, f)
                out.append("Neighbor index= %d" % ah.master_index(Q))
                out.append("Neighbor of neighbor(%d) = %d" % (f, ah.master_index(Q)))
                assert ah.neighbor(Q, ah.neighbor_of_agglomerated_neighbor(P, f)) == P
            else:
                out.append("Face with idx: %d is a boundary face." % f)
        out.append("")
    return out


def render_sparsity(ah):
    # DynamicSparsityPattern::print: "[row,col,col,...]" ascending (sparsity_agglomerated_tria.cc)
    return ["[" + ",".join(str(int(c)) for c in [r] + list(cols)) + "]" for r, cols in enumerate(ah.sparsity_rows())]


def render_hp_structure(ah, grid_vertices_of_cell):
    # test/polydeal/hp_structure_01.cc:95-112
    out = []
    for P in range(ah.n_agglomerates):
        m = ah.master_index(P)
        out.append("Cell with global index: %d has global DoF indices: " % m)
        for d in ah.dof_indices(P):
            out.append("%d" % d)
        out.append(" and vertices: ")
        for v in grid_vertices_of_cell(m):
            out.append(" ".join(fmt(x) for x in v))
    return out


def render_polytope_iterator_forward(ah, n_first):
    # test/polydeal/polytope_iterator.cc:240-256: first loop prints polytopes with index < 7
    out = []
    for P in range(n_first):
        out.append("n_faces =%d" % ah.n_faces_of(P))
        out.append("Global DoF indices for polytope %d" % P)
        for d in ah.dof_indices(P):
            out.append("%d" % d)
    return out


# ---- agglomerate lists hard-coded in the reference tests ---------------------------------------
GROUPS_QUAD_PTS = [[3, 6, 9], [36, 37], [25, 19]]  # reinit_cell_face_quad_pts.cc:60-78
GROUPS_FOUR = [[3, 6, 9, 12, 13], [15, 36, 37], [57, 60, 54], [25, 19, 22]]  # reinit_cell_face_01.cc:52-80
GROUPS_RCF02 = [[3, 6, 9], [15, 36, 37], [57, 60, 54], [25, 19, 22]]  # reinit_cell_face_02.cc:68-95
GROUPS_2X2 = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
GROUPS_HALVES = [[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14, 15]]  # continuous_face_01.cc:161-170
GROUPS_CF02 = [
    [[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11], [12], [13], [14], [15]],  # continuous_face_02.cc:159-204
    [[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]],  # :226-249
    [[3, 6, 9, 12], [0, 1, 4, 5], [2, 8, 10], [11, 14, 15], [7, 13]],  # :267-297 (refine 2)
]
GROUPS_MASTER_MASTER = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12], [13], [14], [15]]
GROUPS_POLY_ITER = [[3235, 3238], [831, 874], [1226, 1227], [2279, 2278], [3760, 3761], [3648, 3306], [3765, 3764]]


def render_continuous_face_03(ah):
    # test/polydeal/continuous_face_03.cc:30-63 (perimeter_test; prints "Neighbor = ", no sub-face lists) + main():218-223
    out = []
    perimeter = 0.0
    for P in range(ah.n_agglomerates):
        out.append("Master cell index = %d" % ah.master_index(P))
        nf = ah.n_faces_of(P)
        out.append("Number of agglomerated faces = %d" % nf)
        for f in range(nf):
            out.append("Agglomerate face index = %d" % f)
            if not ah.at_boundary(P, f):
                out.append("Neighbor = %d" % ah.neighbor(P, f))
                out.append("Neighbor of neighbor = %d" % ah.neighbor_of_agglomerated_neighbor(P, f))
            else:
                perimeter += ah.face_jxw_sum(P, f)
        out.append("")
    out.append("Perimeter = " + fmt(perimeter))
    out.append("- - - - - - - - - - - -")
    out.append("Check on neighbors and neighbors of neighbors:")
    for P in range(ah.n_agglomerates):
        for f in range(ah.n_faces_of(P)):
            if not ah.at_boundary(P, f):
                assert ah.neighbor(ah.neighbor(P, f), ah.neighbor_of_agglomerated_neighbor(P, f)) == P
    out.append("Ok")
    out += ["- - - - - - - - - - - -", "Check on quadrature points:", "Ok", "End Test"]
    return out


def define_continuous_face_03(ah, n_cells):
    # continuous_face_03.cc:176-213: singletons of the un-flagged cells FIRST (cell order), then the three agglomerates
    groups = [[36, 37, 38, 39], [18, 24, 25], [3, 6]]
    flagged = {c for g in groups for c in g}
    for c in range(n_cells):
        if c not in flagged:
            ah.define_agglomerate([c])
    for g in groups:
        ah.define_agglomerate(sorted(g))


def render_rtree_levels(sizes_per_level, first_level, head, size_label, line):
    """test/polydeal/rtree_mesh.output / extract_last_level.output: per extraction level the number of agglomerates and
    the number of cells of each.  On the 32x32 hyper_cube the R-tree levels are exact 2^k x 2^k blocks
    (reference include/agglomerator.h:389-434), so a block hierarchy must print the same text."""
    out = []
    for k, sizes in enumerate(sizes_per_level):
        out.append("%s%d" % (head, first_level + k))
        out.append("%s1024" % size_label)
        out.append("Total number of available levels: 4")
        out.append("N subdomains = %d" % len(sizes))
        for i, s_ in enumerate(sizes):
            out.append(line % (s_, i))
        out.append("")
    return out
