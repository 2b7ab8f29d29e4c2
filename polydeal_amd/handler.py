"""Python face of the C++ host mirror (polydeal_amd/csrc/host/polydeal_host.h) — same names as the
reference's operator surface for this path: AgglomerationHandler, FE_DGQ, FE_AggloDGP,
assemble_dg_matrix (reference include/agglomeration_handler.h:203-452, include/fe_agglodgp.h:317,
include/poly_utils.h:2000-2195).  All logic lives in C++; this file only marshals."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import PDH_BASIS_AGGLODGP, PDH_BASIS_DGQ, Context, PdhError, pdh_problem

_ready = False


def _lib():
    global _ready
    lib = _capi.load_library()
    if not _ready:
        lib.pdhh_last_error.restype = C.c_char_p
        lib.pdhh_grid_create.restype = C.c_void_p
        lib.pdhh_grid_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        lib.pdhh_grid_create_rectangle.restype = C.c_void_p
        lib.pdhh_grid_create_rectangle.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pdhh_grid_destroy.argtypes = [C.c_void_p]
        lib.pdhh_grid_destroy.restype = None
        lib.pdhh_grid_n_cells.argtypes = [C.c_void_p]
        lib.pdhh_grid_distort.argtypes = [C.c_void_p, C.c_double, C.c_uint]
        lib.pdhh_blocks_per_row.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdhh_flatten_cartesian.restype = C.c_void_p
        lib.pdhh_flatten_cartesian.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_int]
        lib.pdhh_flat_cartesian.restype = C.c_void_p
        lib.pdhh_flat_cartesian.argtypes = [C.c_void_p]
        lib.pdhh_grid_vertices.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.pdhh_handler_create.restype = C.c_void_p
        lib.pdhh_handler_create.argtypes = [C.c_void_p]
        lib.pdhh_handler_destroy.argtypes = [C.c_void_p]
        lib.pdhh_handler_destroy.restype = None
        lib.pdhh_define_agglomerate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.pdhh_define_block_agglomerates.argtypes = [C.c_void_p, C.c_int]
        lib.pdhh_grid_read_msh.argtypes = [C.c_char_p, C.c_int]
        lib.pdhh_grid_read_msh.restype = C.c_void_p
        lib.pdhh_grid_neighbor.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pdhh_define_grown_agglomerates.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        lib.pdhh_get_agglomerate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.pdhh_partition_into_grown_agglomerates.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        lib.pdhh_initialize_fe_values.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pdhh_distribute_agglomerated_dofs.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for name in ("pdhh_n_agglomerates", "pdhh_n_dofs", "pdhh_n_dofs_per_cell"):
            getattr(lib, name).argtypes = [C.c_void_p]
        for name in ("pdhh_master_index", "pdhh_master_slave_value", "pdhh_n_faces", "pdhh_dof_offset", "pdhh_agglomerate_size"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
        for name in ("pdhh_at_boundary", "pdhh_neighbor", "pdhh_neighbor_of_agglomerated_neighbor"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pdhh_interface.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.pdhh_bbox.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.pdhh_diameter.argtypes = [C.c_void_p, C.c_int]
        lib.pdhh_diameter.restype = C.c_double
        lib.pdhh_volume_jxw_sum.argtypes = [C.c_void_p, C.c_int]
        lib.pdhh_volume_jxw_sum.restype = C.c_double
        lib.pdhh_face_jxw_sum.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pdhh_face_jxw_sum.restype = C.c_double
        lib.pdhh_sparsity.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.pdhh_sparsity.restype = C.c_int64
        lib.pdhh_flatten.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        lib.pdhh_flatten.restype = C.c_void_p
        lib.pdhh_flatten_local.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        lib.pdhh_flatten_local.restype = C.c_void_p
        lib.pdhh_flat_problem.argtypes = [C.c_void_p]
        lib.pdhh_flat_problem.restype = C.POINTER(pdh_problem)
        lib.pdhh_flat_destroy.argtypes = [C.c_void_p]
        lib.pdhh_flat_destroy.restype = None
        lib.pdhh_flat_local_of.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdhh_flat_sizes.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdhh_flat_sizes.restype = C.c_int64
        lib.pdhh_assemble_dg_matrix.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                                                C.c_int, C.c_int, C.c_void_p, C.c_int64]
        lib.pdhh_fill_injection_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_int64]
        _ready = True
    return lib


class HostError(RuntimeError):
    pass


def _raise():
    raise HostError(_lib().pdhh_last_error().decode())


@dataclass(frozen=True)
class FiniteElement:
    dim: int
    degree: int
    basis: int

    @property
    def n_dofs_per_cell(self):
        import math
        if self.basis == PDH_BASIS_DGQ:
            return (self.degree + 1) ** self.dim
        return math.comb(self.degree + self.dim, self.dim)


def FE_DGQ(dim, degree):
    return FiniteElement(dim, degree, PDH_BASIS_DGQ)


def FE_AggloDGP(dim, degree):
    return FiniteElement(dim, degree, PDH_BASIS_AGGLODGP)


@dataclass
class SipVariant:
    """Scalars that differ between the reference's callers (SURVEY.md 8(a) 'Variants')."""
    penalty_constant: float = -1.0  # <0: 10 (p+dim)(p+1)   poly_utils.h:2018-2019
    owner_rule: int = 0  # 0: id()<id() (poly_utils.h:2089), 1: index()<index() (poisson.cc:841)
    h_rule: int = 0  # 0: C/diameter(owner), 1: C (h_f=1), 2: C max(1/h_in,1/h_out) (minimal_SIP.cc:256-259)
    boundary: int = 0  # 0: Nitsche, 1: zeroed (minimal_SIP.cc:230-248)
    reaction_c: float = 0.0  # diffusion_reaction.cc:495-501

    @staticmethod
    def assemble_dg_matrix():
        return SipVariant()

    @staticmethod
    def poisson_example(fe):
        return SipVariant(10.0 * (fe.degree + 1) * (fe.degree + fe.dim), 1)

    @staticmethod
    def minimal_sip_test():
        return SipVariant(20.0, 1, 1)

    @staticmethod
    def minimal_sip_example():
        return SipVariant(10.0, 1, 2, 1)

    @staticmethod
    def diffusion_reaction(fe):
        return SipVariant(10.0 * fe.degree ** 2, 0, 0, 0, 0.5)


class BackgroundGrid:
    """hyper_cube + refine_global (Morton order) or subdivided_hyper_cube (lexicographic)."""

    def __init__(self, dim, n_per_dir, lo=0.0, hi=1.0, morton=True, _handle=None):
        self.dim, self.n_per_dir = dim, n_per_dir
        self.h = _handle if _handle is not None else _lib().pdhh_grid_create(dim, n_per_dir, int(morton), lo, hi)
        if not self.h:
            _raise()

    @staticmethod
    def subdivided_hyper_rectangle(dim, repetitions, lo, hi):
        """GridGenerator::subdivided_hyper_rectangle(tria, repetitions, p1, p2): lexicographic cells."""
        rep = np.ascontiguousarray(repetitions, dtype=np.int32)
        p1 = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.float64), (dim,)))
        p2 = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.float64), (dim,)))
        if rep.shape != (dim,):
            raise ValueError("repetitions must have dim entries")
        h = _lib().pdhh_grid_create_rectangle(dim, rep.ctypes.data, p1.ctypes.data, p2.ctypes.data)
        if not h:
            _raise()
        g = BackgroundGrid(dim, int(rep[0]), _handle=h)
        g.repetitions = tuple(int(r) for r in rep)
        return g

    @staticmethod
    def hyper_cube_refined(dim, lo, hi, n_refine):
        return BackgroundGrid(dim, 2 ** n_refine, lo, hi, True)

    @staticmethod
    def read_msh(path, n_refine=0):
        """GridIn<2>::read_msh (gmsh 4.1 ASCII, quadrilaterals) + Triangulation::refine_global(n_refine): an unstructured
        background mesh (reference examples/minimal_SIP.cc:94-118 reads meshes/t3.msh this way)."""
        h = _lib().pdhh_grid_read_msh(str(path).encode(), int(n_refine))
        if not h:
            _raise()
        return BackgroundGrid(2, 0, _handle=h)

    def neighbor(self, cell, f):
        """cell->neighbor(f), -1 on the domain boundary."""
        return _lib().pdhh_grid_neighbor(self.h, int(cell), int(f))

    @staticmethod
    def subdivided_hyper_cube(dim, n, lo=0.0, hi=1.0):
        return BackgroundGrid(dim, n, lo, hi, False)

    @property
    def n_cells(self):
        return _lib().pdhh_grid_n_cells(self.h)

    def distort(self, factor, seed=0):
        if _lib().pdhh_grid_distort(self.h, factor, seed) < 0:
            _raise()
        return self

    def cell_vertices(self, cell):
        out = np.zeros((2 ** self.dim, self.dim))
        if _lib().pdhh_grid_vertices(self.h, cell, out.ctypes.data) < 0:
            _raise()
        return out

    def __del__(self):
        try:
            if self.h:
                _lib().pdhh_grid_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _FlatOwner:
    """Owns one flattened problem of the C++ host mirror (independent of the handler and of later flatten() calls)."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                _lib().pdhh_flat_destroy(self.h)
                self.h = None
        except Exception:
            pass


class FlatView:
    """A flattened problem (pdh_problem + its arrays) produced by AgglomerationHandler.flatten / flatten_local.
    Every view owns its description; the NumPy arrays returned by arrays() keep it alive."""

    def __init__(self, fh):
        self._own = _FlatOwner(fh)
        self.c = _lib().pdhh_flat_problem(fh).contents
        sizes = (C.c_int64 * 4)()
        _lib().pdhh_flat_sizes(fh, sizes)
        self.nq_tot, self.nqf_tot, self.nnz, self.n_faces = [int(x) for x in sizes]
        self.local = bool(self.c.local)
        self.cartesian = _lib().pdhh_flat_cartesian(fh)  # address of the pdh_cartesian_points of a flatten_cartesian view, or None

    @property
    def n_local_rows(self):
        """Rows covered by rowptr: all of them for a global description, the owned ones for a rank-local one."""
        return self._n_local_rows if self.local else self.c.n_rows

    def local_of(self):
        """Global polytope index of every local polytope (rank-local descriptions)."""
        out = np.zeros(self.c.n_agg, dtype=np.int32)
        n = _lib().pdhh_flat_local_of(self._own.h, out.ctypes.data)
        return out[:n]

    def _arr(self, field, ctype, n):
        p = getattr(self.c, field)
        if not p or n == 0:
            return None
        buf = (ctype * n).from_address(p)
        buf._owner = self._own  # the array's base chain ends here: the C++ storage outlives every view of it
        return np.frombuffer(buf, dtype=np.dtype(ctype))

    def arrays(self):
        c, d = self.c, self.c.dim
        return dict(
            bbox=self._arr("bbox", C.c_double, c.n_agg * 2 * d), dof_offset=self._arr("dof_offset", C.c_int32, c.n_agg),
            vq_ptr=self._arr("vq_ptr", C.c_int64, c.n_agg + 1), vq_x=self._arr("vq_x", C.c_double, d * self.nq_tot),
            vq_w=self._arr("vq_w", C.c_double, self.nq_tot),
            face_in=self._arr("face_in", C.c_int32, c.n_faces), face_out=self._arr("face_out", C.c_int32, c.n_faces),
            fq_ptr=self._arr("fq_ptr", C.c_int64, c.n_faces + 1), fq_x=self._arr("fq_x", C.c_double, d * self.nqf_tot),
            fq_n=self._arr("fq_n", C.c_double, d * self.nqf_tot), fq_w=self._arr("fq_w", C.c_double, self.nqf_tot),
            fq_w_out=self._arr("fq_w_out", C.c_double, self.nqf_tot),
            face_sigma=self._arr("face_sigma", C.c_double, c.n_faces),
            rowptr=self._arr("rowptr", C.c_int64, self.n_local_rows + 1), colind=self._arr("colind", C.c_int32, self.nnz),
            col_offset=self._arr("col_offset", C.c_int32, c.n_agg), agg_rank=self._arr("agg_rank", C.c_int32, c.n_agg),
        )


class AgglomerationHandler:
    """Mirror of AgglomerationHandler<dim> restricted to the SIP path (polytopes addressed by index())."""

    def __init__(self, grid: BackgroundGrid):
        self.grid = grid
        self.h = _lib().pdhh_handler_create(grid.h)
        self.fe = None

    # -- reference API -------------------------------------------------------------------------------
    def define_agglomerate(self, cells):
        a = np.ascontiguousarray(cells, dtype=np.int32)
        r = _lib().pdhh_define_agglomerate(self.h, a.ctypes.data, len(a))
        if r < 0:
            _raise()
        return r

    def define_block_agglomerates(self, b):
        if _lib().pdhh_define_block_agglomerates(self.h, b) < 0:
            _raise()

    def define_grown_agglomerates(self, cells_per_polytope, seed=0):
        """Connected irregular agglomerates grown over the cell graph (stand-in for the METIS partition of
        reference examples/poisson.cc:543-566)."""
        if _lib().pdhh_define_grown_agglomerates(self.h, int(cells_per_polytope), int(seed)) < 0:
            _raise()

    def partition_into_grown_agglomerates(self, n_subdomains, seed=0):
        """n_subdomains connected agglomerates (stand-in for GridTools::partition_triangulation(n, tria, metis) + one
        agglomerate per subdomain, reference examples/minimal_SIP.cc:107-139)."""
        if _lib().pdhh_partition_into_grown_agglomerates(self.h, int(n_subdomains), int(seed)) < 0:
            _raise()

    def get_agglomerate(self, P):
        """Cells of polytope P: slaves in insertion order, then the master (reference include/agglomeration_handler.h:1022-1032)."""
        cap = 64
        while True:
            a = np.zeros(cap, dtype=np.int32)
            n = _lib().pdhh_get_agglomerate(self.h, int(P), a.ctypes.data, cap)
            if n < 0:
                _raise()
            if n <= cap:
                return a[:n].tolist()
            cap = n

    def initialize_fe_values(self, n_q_points_1d, n_face_q_points_1d):
        if _lib().pdhh_initialize_fe_values(self.h, n_q_points_1d, n_face_q_points_1d) < 0:
            _raise()

    def distribute_agglomerated_dofs(self, fe: FiniteElement):
        if _lib().pdhh_distribute_agglomerated_dofs(self.h, fe.basis, fe.degree) < 0:
            _raise()
        self.fe = fe

    @property
    def n_agglomerates(self):
        return _lib().pdhh_n_agglomerates(self.h)

    @property
    def n_dofs(self):
        return _lib().pdhh_n_dofs(self.h)

    @property
    def n_dofs_per_cell(self):
        return _lib().pdhh_n_dofs_per_cell(self.h)

    # -- accessor-level queries (shared protocol with the oracle, see tests/golden_cases.py) -----------
    def master_index(self, P):
        return _lib().pdhh_master_index(self.h, P)

    def master_slave_value(self, cell):
        return _lib().pdhh_master_slave_value(self.h, cell)

    def agglomerate_size(self, P):
        """Number of cells of polytope P (polytope->get_agglomerate().size())."""
        return _lib().pdhh_agglomerate_size(self.h, P)

    def blocks_per_row(self):
        """1 + number of neighbours of every polytope, in dof order: weights for partition.balanced_row_splits."""
        out = np.zeros(self.n_agglomerates, dtype=np.int32)
        if _lib().pdhh_blocks_per_row(self.h, out.ctypes.data) < 0:
            _raise()
        return out

    def n_faces_of(self, P):
        return _lib().pdhh_n_faces(self.h, P)

    def at_boundary(self, P, f):
        return bool(_lib().pdhh_at_boundary(self.h, P, f))

    def neighbor(self, P, f):
        r = _lib().pdhh_neighbor(self.h, P, f)
        if r == -2:
            _raise()
        return r

    def neighbor_of_agglomerated_neighbor(self, P, f):
        r = _lib().pdhh_neighbor_of_agglomerated_neighbor(self.h, P, f)
        if r == -2:
            _raise()
        return r

    def interface_list(self, P, Q):
        cap = 4096
        cells = np.zeros(cap, dtype=np.int32)
        faces = np.zeros(cap, dtype=np.int32)
        n = _lib().pdhh_interface(self.h, P, Q, cells.ctypes.data, faces.ctypes.data, cap)
        if n < 0:
            _raise()
        return [(int(cells[i]), int(faces[i])) for i in range(min(n, cap))]

    def bbox(self, P):
        d = self.grid.dim
        out = np.zeros(2 * d)
        if _lib().pdhh_bbox(self.h, P, out.ctypes.data) < 0:
            _raise()
        return out[:d], out[d:]

    def diameter(self, P):
        return _lib().pdhh_diameter(self.h, P)

    def dof_indices(self, P):
        off = _lib().pdhh_dof_offset(self.h, P)
        return np.arange(off, off + self.n_dofs_per_cell)

    def volume_jxw_sum(self, P):
        return _lib().pdhh_volume_jxw_sum(self.h, P)

    def face_jxw_sum(self, P, f):
        return _lib().pdhh_face_jxw_sum(self.h, P, f)

    def sparsity_pattern(self, diag_first=True):
        """create_agglomeration_sparsity_pattern -> (rowptr, colind)."""
        rowptr = np.zeros(self.n_dofs + 1, dtype=np.int64)
        nnz = _lib().pdhh_sparsity(self.h, int(diag_first), rowptr.ctypes.data, None)
        if nnz < 0:
            _raise()
        colind = np.zeros(nnz, dtype=np.int32)
        _lib().pdhh_sparsity(self.h, int(diag_first), rowptr.ctypes.data, colind.ctypes.data)
        return rowptr, colind

    def sparsity_rows(self):
        rp, ci = self.sparsity_pattern(diag_first=False)
        return [ci[rp[r]:rp[r + 1]] for r in range(self.n_dofs)]

    # -- flattening + assembly -----------------------------------------------------------------------
    def flatten(self, variant: SipVariant, diag_first=True, with_colind=False) -> FlatView:
        p = _lib().pdhh_flatten(self.h, variant.penalty_constant, variant.owner_rule, variant.h_rule,
                                variant.boundary, variant.reaction_c, int(diag_first), int(with_colind))
        if not p:
            _raise()
        return FlatView(p)

    def flatten_cartesian(self, variant: SipVariant, diag_first=True, with_colind=False, row_begin=0, row_end=0, row_splits=None) -> FlatView:
        """The description WITHOUT the points (agglomerates of Cartesian cells, 3-D): every group of quadrature points is named by its
        cell (and local face); Context.set_problem takes the view and has the points generated on the device
        (pdh_set_problem_cartesian).  row_end > row_begin: rank-local description of those rows."""
        rs = None if row_splits is None else np.ascontiguousarray(row_splits, dtype=np.int32)
        p = _lib().pdhh_flatten_cartesian(self.h, variant.penalty_constant, variant.owner_rule, variant.h_rule, variant.boundary,
                                          variant.reaction_c, int(diag_first), int(with_colind), int(row_begin), int(row_end),
                                          None if rs is None else rs.ctypes.data, 0 if rs is None else len(rs) - 1)
        if not p:
            _raise()
        v = FlatView(p)
        if row_end > row_begin:
            v._n_local_rows = row_end - row_begin
        return v

    def flatten_local(self, variant: SipVariant, row_begin, row_end, diag_first=True, with_colind=False, row_splits=None,
                      epetra_columns=False) -> FlatView:
        """Rank-local description (pdh_problem.local = 1) of the dof rows [row_begin,row_end): the owned polytopes plus
        their ghost neighbours, global dof numbers, rowptr/colind of the owned rows only - what one MPI rank of the
        reference holds (source/agglomeration_handler.cc:1026-1091).  row_splits [n_ranks+1] (first row of every rank)
        fills agg_rank; epetra_columns orders rows by Epetra local column ids (owned first, ghosts behind)."""
        rs = None if row_splits is None else np.ascontiguousarray(row_splits, dtype=np.int32)
        p = _lib().pdhh_flatten_local(self.h, variant.penalty_constant, variant.owner_rule, variant.h_rule, variant.boundary,
                                      variant.reaction_c, int(diag_first), int(with_colind), int(row_begin), int(row_end),
                                      None if rs is None else rs.ctypes.data, 0 if rs is None else len(rs) - 1,
                                      int(epetra_columns))
        if not p:
            _raise()
        fv = FlatView(p)
        fv._n_local_rows = int(row_end) - int(row_begin)
        return fv

    def __del__(self):
        try:
            if self.h:
                _lib().pdhh_handler_destroy(self.h)
                self.h = None
        except Exception:
            pass


def assemble_dg_matrix(fe: FiniteElement, ah: AgglomerationHandler, variant: SipVariant | None = None,
                       diag_first=True, device=0):
    """PolyUtils::assemble_dg_matrix (include/poly_utils.h:2000-2195) on the GPU.
    Returns (rowptr, colind, values) of the pattern the handler creates.  Raises if no HIP device."""
    if fe != ah.fe:
        raise ValueError("FE passed to assemble_dg_matrix differs from the handler's")
    variant = variant or SipVariant.assemble_dg_matrix()
    rowptr, colind = ah.sparsity_pattern(diag_first)
    values = np.zeros(int(rowptr[-1]))
    rc = _lib().pdhh_assemble_dg_matrix(ah.h, variant.penalty_constant, variant.owner_rule, variant.h_rule,
                                         variant.boundary, variant.reaction_c, int(diag_first), device,
                                         values.ctypes.data, len(values))
    if rc < 0:
        _raise()
    return rowptr, colind, values


def fill_injection_matrix(coarse_ah: AgglomerationHandler, fine_ah: AgglomerationHandler, device=0):
    """Utils::fill_injection_matrix (reference include/utils.h:95-270): CSR (rowptr, colind, values) of the
    injection from the coarse polytopal space into the fine one; basis evaluation on the GPU
    (pdh_shape_values).  The handlers must be nested over the same grid and carry the same FE_DGQ."""
    if coarse_ah.grid is not fine_ah.grid:
        raise ValueError("both handlers must live on the same grid")
    n = fine_ah.n_dofs_per_cell
    rows = fine_ah.n_dofs
    rowptr = np.zeros(rows + 1, dtype=np.int64)
    colind = np.zeros(rows * n, dtype=np.int32)
    values = np.zeros(rows * n)
    rc = _lib().pdhh_fill_injection_matrix(coarse_ah.h, fine_ah.h, device, rowptr.ctypes.data, colind.ctypes.data,
                                            values.ctypes.data, len(values))
    if rc < 0:
        _raise()
    return rowptr, colind, values
