"""The compact description of agglomerates of Cartesian cells (flatten_cartesian -> pdh_set_problem_cartesian) on the CPU: it must
name exactly the quadrature data the points-based description carries.  The closed forms the device generator uses
(polydeal_amd/csrc/pdh_cartgen.hip) are restated here in NumPy and compared with the points the host mirror gathers cell by cell
(reference source/agglomeration_handler.cc:622-707, 1146-1165)."""
import ctypes as C

import numpy as np
import pytest

import polydeal_amd as pa


def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _cart_arrays(view):
    cp = C.cast(view.cartesian, C.POINTER(_Cart)).contents
    nsv = view.nq_tot // cp.nq ** 3
    nsf = view.nqf_tot // cp.nqf ** 2
    box = np.ctypeslib.as_array(C.cast(cp.cell_box, C.POINTER(C.c_double)), (cp.n_cells, 6)).copy()
    vq_cell = np.ctypeslib.as_array(C.cast(cp.vq_cell, C.POINTER(C.c_int32)), (nsv,)).copy()
    fq_cell = np.ctypeslib.as_array(C.cast(cp.fq_cell, C.POINTER(C.c_int32)), (max(nsf, 1),)).copy()[:nsf]
    fq_face = np.ctypeslib.as_array(C.cast(cp.fq_face, C.POINTER(C.c_int32)), (max(nsf, 1),)).copy()[:nsf]
    return cp.nq, cp.nqf, box, vq_cell, fq_cell, fq_face


class _Cart(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("cell_box", C.c_void_p), ("vq_cell", C.c_void_p), ("fq_cell", C.c_void_p), ("fq_face", C.c_void_p),
                ("nq", C.c_int32), ("nqf", C.c_int32)]


@pytest.mark.parametrize("kind,cells,per,nq", [("block", 4, 2, 3), ("grown", 6, 5, 4), ("rect", 4, 2, 2)])
def test_compact_description_names_the_points_of_the_full_one(kind, cells, per, nq):
    if kind == "rect":
        grid = pa.BackgroundGrid.subdivided_hyper_rectangle(3, (4, 2, 6), (0.0, -1.0, 0.5), (1.0, 0.0, 2.0))
    else:
        grid = pa.BackgroundGrid.subdivided_hyper_cube(3, cells, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    if kind == "grown":
        ah.define_grown_agglomerates(per, seed=3)
    else:
        ah.define_block_agglomerates(per)
    fe = pa.FE_AggloDGP(3, 2)
    ah.initialize_fe_values(nq, nq)
    ah.distribute_agglomerated_dofs(fe)
    var = pa.SipVariant.poisson_example(fe)
    full, comp = ah.flatten(var, True, True), ah.flatten_cartesian(var, True, True)
    fa, ca = full.arrays(), comp.arrays()
    for key in ("bbox", "dof_offset", "vq_ptr", "face_in", "face_out", "fq_ptr", "face_sigma", "rowptr", "colind"):
        assert np.array_equal(fa[key], ca[key]), key
    for key in ("vq_x", "vq_w", "fq_x", "fq_n", "fq_w"):
        assert ca[key] is None
    n1, n2, box, vq_cell, fq_cell, fq_face = _cart_arrays(comp)
    assert (n1, n2) == (nq, nq)
    x1, w1 = gauss01(nq)
    # volume groups: x fastest
    N = full.nq_tot
    vx, vw = fa["vq_x"].reshape(3, N), fa["vq_w"]
    idx = np.arange(nq ** 3)
    i = np.stack([idx % nq, (idx // nq) % nq, idx // (nq * nq)])
    for g in np.unique(np.linspace(0, len(vq_cell) - 1, 40).astype(int)):
        b = box[vq_cell[g]]
        h = b[3:] - b[:3]
        pts = b[:3, None] + h[:, None] * x1[i]
        w = np.prod(h) * w1[i[0]] * w1[i[1]] * w1[i[2]]
        sl = slice(g * nq ** 3, (g + 1) * nq ** 3)
        assert np.max(np.abs(vx[:, sl] - pts)) <= 4e-16 * (1 + np.max(np.abs(pts)))
        assert np.max(np.abs(vw[sl] - w)) <= 1e-14 * np.max(w)
    # face groups: the generator runs the lower tangential axis fastest; the mirror's order differs on y-faces - compare as sets
    M = full.nqf_tot
    fx, fn, fw = fa["fq_x"].reshape(3, M), fa["fq_n"].reshape(3, M), fa["fq_w"]
    j = np.arange(nq * nq)
    for s in np.unique(np.linspace(0, len(fq_cell) - 1, 60).astype(int)):
        b, f = box[fq_cell[s]], int(fq_face[s])
        c, side = f >> 1, f & 1
        ti, tj = (1 if c == 0 else 0), (1 if c == 2 else 2)
        h = b[3:] - b[:3]
        pts = np.zeros((3, nq * nq))
        pts[c] = b[3 + c] if side else b[c]
        pts[ti] = b[ti] + h[ti] * x1[j % nq]
        pts[tj] = b[tj] + h[tj] * x1[j // nq]
        w = h[ti] * h[tj] * w1[j % nq] * w1[j // nq]
        sl = slice(s * nq * nq, (s + 1) * nq * nq)
        key_g = np.lexsort(np.round(pts, 12))
        key_f = np.lexsort(np.round(fx[:, sl], 12))
        assert np.max(np.abs(fx[:, sl][:, key_f] - pts[:, key_g])) <= 4e-16 * (1 + np.max(np.abs(pts)))
        assert np.max(np.abs(fw[sl][key_f] - w[key_g])) <= 1e-14 * np.max(w)
        nrm = np.zeros(3)
        nrm[c] = 1.0 if side else -1.0
        assert np.max(np.abs(fn[:, sl] - nrm[:, None])) <= 1e-15


def test_compact_description_is_small_and_refuses_distorted_cells():
    grid = pa.BackgroundGrid.subdivided_hyper_cube(3, 8, 0.0, 1.0)
    ah = pa.AgglomerationHandler(grid)
    ah.define_block_agglomerates(2)
    fe = pa.FE_DGQ(3, 3)
    ah.initialize_fe_values(4, 4)
    ah.distribute_agglomerated_dofs(fe)
    comp = ah.flatten_cartesian(pa.SipVariant.poisson_example(fe))
    n1, n2, box, vq_cell, fq_cell, fq_face = _cart_arrays(comp)
    assert len(vq_cell) == 512 and box.shape == (512, 6) and len(fq_cell) == comp.nqf_tot // 16
    assert sorted(vq_cell.tolist()) == list(range(512))  # every cell in exactly one polytope
    grid2 = pa.BackgroundGrid.subdivided_hyper_cube(3, 4, 0.0, 1.0)
    grid2.distort(0.05, 2)
    ah2 = pa.AgglomerationHandler(grid2)
    ah2.define_block_agglomerates(2)
    ah2.initialize_fe_values(4, 4)
    ah2.distribute_agglomerated_dofs(fe)
    with pytest.raises(Exception, match="box"):
        ah2.flatten_cartesian(pa.SipVariant.poisson_example(fe))
    g2d = pa.BackgroundGrid.subdivided_hyper_cube(2, 4, 0.0, 1.0)
    ah3 = pa.AgglomerationHandler(g2d)
    ah3.define_block_agglomerates(2)
    ah3.initialize_fe_values(3, 3)
    ah3.distribute_agglomerated_dofs(pa.FE_DGQ(2, 2))
    with pytest.raises(Exception, match="3-D"):
        ah3.flatten_cartesian(pa.SipVariant.poisson_example(pa.FE_DGQ(2, 2)))
