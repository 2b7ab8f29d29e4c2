"""CPU restatement (NumPy) of polyDEAL's SIP assembly over agglomerated polytopes.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``polydeal_amd``) never
imports this module and fails loudly when its HIP library is missing.

Parity status: *pinned* for the mesh numbering, master/slave bookkeeping, bounding
boxes, face enumeration, neighbour tables, DoF numbering, sparsity rows, volume and
face quadrature sums against the golden ``.output`` files the reference's own tests
hold (see ``tests/golden/`` and ``tests/test_oracle_golden.py``), and for the assembled
operator through the reference's known-answer identities (agglomerated == standard SIP
entry by entry, v^T A v identities, exact-solution reproduction).  *Unpinned* (no
reference fixture exists; deal.II is not installed here and the reference cannot be
built): intra-block basis order of FE_DGQ/FE_AggloDGP for p >= 2, Gauss-Lobatto node
placement for p >= 3 and the order of quadrature points inside a face.  Those follow
deal.II's documented conventions and are marked [deal.II] below.

Every function cites the reference file:line it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field

import numpy as np

INVALID = -1
BDRY_KEY = np.iinfo(np.uint32).max  # source/agglomeration_handler.cc:1575-1577 (UINT_MAX key)


# ---------------------------------------------------------------------------
# 1-D rules and bases  [deal.II conventions, SURVEY.md 8(c) items 3-5]
# ---------------------------------------------------------------------------
def qgauss_1d(n: int):
    """QGauss<1>(n): n-point Gauss-Legendre rule on [0,1]  [deal.II]."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_lobatto_nodes(p: int) -> np.ndarray:
    """The p+1 Gauss-Lobatto points on [0,1] (support points of FE_DGQ(p))  [deal.II]."""
    if p == 0:
        return np.array([0.5])
    if p == 1:
        return np.array([0.0, 1.0])
    inner = np.polynomial.legendre.Legendre.basis(p).deriv().roots()
    inner = np.sort(np.real(inner))
    # Newton polish on P_p'(x)
    P = np.polynomial.legendre.Legendre.basis(p)
    d1, d2 = P.deriv(1), P.deriv(2)
    for _ in range(3):
        inner = inner - d1(inner) / d2(inner)
    x = np.concatenate([[-1.0], inner, [1.0]])
    return 0.5 * (x + 1.0)


def lagrange_1d(nodes: np.ndarray, x: np.ndarray):
    """Values and first derivatives of the Lagrange basis on `nodes` at points x.

    Returns (val[k, q], der[k, q]).  Product form, as deal.II's
    Polynomials::generate_complete_Lagrange_basis evaluates it  [deal.II]."""
    x = np.asarray(x, dtype=np.float64)
    m = len(nodes)
    val = np.ones((m, x.size))
    der = np.zeros((m, x.size))
    for k in range(m):
        denom = 1.0
        for j in range(m):
            if j != k:
                denom *= nodes[k] - nodes[j]
        for j in range(m):
            if j == k:
                continue
            val[k] *= x - nodes[j]
            term = np.ones(x.size)
            for l in range(m):
                if l != k and l != j:
                    term = term * (x - nodes[l])
            der[k] += term
        val[k] /= denom
        der[k] /= denom
    return val, der


def legendre_1d(p: int, x: np.ndarray):
    """Polynomials::Legendre(k)(x) = sqrt(2k+1) P_k(2x-1), k=0..p, with derivatives.

    L2-orthonormal on [0,1]  [deal.II]; used by FE_AggloDGP
    (source/fe_agglodgp.cc:27-31)."""
    x = np.asarray(x, dtype=np.float64)
    t = 2.0 * x - 1.0
    val = np.zeros((p + 1, x.size))
    der = np.zeros((p + 1, x.size))
    val[0] = 1.0
    if p >= 1:
        val[1] = t
        der[1] = 1.0
    for k in range(1, p):
        val[k + 1] = ((2 * k + 1) * t * val[k] - k * val[k - 1]) / (k + 1)
        der[k + 1] = ((2 * k + 1) * (val[k] + t * der[k]) - k * der[k - 1]) / (k + 1)
    for k in range(p + 1):
        s = math.sqrt(2 * k + 1)
        val[k] *= s
        der[k] *= 2.0 * s  # d/dx = 2 d/dt
    return val, der


# ---------------------------------------------------------------------------
# Finite elements on the unit box
# ---------------------------------------------------------------------------
class FE_DGQ:
    """FE_DGQ<dim>(p): tensor Lagrange on Gauss-Lobatto nodes, lexicographic, x fastest
    [deal.II]; n = (p+1)^dim."""

    name = "FE_DGQ"
    basis_id = 0

    def __init__(self, dim: int, degree: int):
        self.dim, self.degree = dim, degree
        self.n1d = degree + 1
        self.nodes = gauss_lobatto_nodes(degree)
        self.multi_index = np.array(
            [idx[::-1] for idx in itertools.product(range(self.n1d), repeat=dim)], dtype=np.int32
        )  # x fastest
        self.n_dofs_per_cell = len(self.multi_index)

    def eval_1d(self, x):
        return lagrange_1d(self.nodes, x)

    def shape(self, unit_pts: np.ndarray):
        """values[q,i], unit-gradients[q,i,c] at unit points [q,c]."""
        return _tensor_shape(self, unit_pts)


class FE_AggloDGP:
    """FE_AggloDGP<dim>(p) (include/fe_agglodgp.h:310-471, source/fe_agglodgp.cc:27-55):
    complete-degree space P_p with tensorised orthonormal Legendre basis;
    PolynomialSpace<dim> index order 'for iz: for iy<n1d-iz: for ix<n1d-iy-iz' (x fastest)
    [deal.II]; n = C(p+dim,dim) (source/fe_agglodgp.cc:89-101)."""

    name = "FE_AggloDGP"
    basis_id = 1

    def __init__(self, dim: int, degree: int):
        self.dim, self.degree = dim, degree
        self.n1d = degree + 1
        mi = []
        if dim == 2:
            for iy in range(self.n1d):
                for ix in range(self.n1d - iy):
                    mi.append((ix, iy))
        elif dim == 3:
            for iz in range(self.n1d):
                for iy in range(self.n1d - iz):
                    for ix in range(self.n1d - iy - iz):
                        mi.append((ix, iy, iz))
        else:
            raise ValueError("dim must be 2 or 3")
        self.multi_index = np.array(mi, dtype=np.int32)
        self.n_dofs_per_cell = len(mi)
        assert self.n_dofs_per_cell == math.comb(degree + dim, dim)

    def eval_1d(self, x):
        return legendre_1d(self.degree, x)

    def shape(self, unit_pts: np.ndarray):
        return _tensor_shape(self, unit_pts)


def _tensor_shape(fe, unit_pts):
    unit_pts = np.atleast_2d(np.asarray(unit_pts, dtype=np.float64))
    nq, dim = unit_pts.shape
    vals1d, ders1d = [], []
    for c in range(dim):
        v, d = fe.eval_1d(unit_pts[:, c])
        vals1d.append(v)
        ders1d.append(d)
    n = fe.n_dofs_per_cell
    val = np.ones((nq, n))
    grad = np.ones((nq, n, dim))
    for c in range(dim):
        k = fe.multi_index[:, c]
        val *= vals1d[c][k].T
        for g in range(dim):
            grad[:, :, g] *= (ders1d[c][k] if g == c else vals1d[c][k]).T
    return val, grad


# ---------------------------------------------------------------------------
# Background grid: hyper_cube + refine_global (Morton) or subdivided (lexicographic)
# ---------------------------------------------------------------------------
def _morton_encode(ijk: np.ndarray, dim: int, levels: int) -> np.ndarray:
    idx = np.zeros(ijk.shape[0], dtype=np.int64)
    for lev in range(levels):
        for c in range(dim):
            idx |= ((ijk[:, c] >> lev) & 1) << (dim * lev + c)
    return idx


@dataclass
class Grid:
    """Quad/hex background mesh with deal.II numbering conventions [deal.II]:
    faces 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z; vertex v at ((v&1),(v>>1)&1,(v>>2)&1);
    `refine_global(k)` of one coarse cell numbers active cells in Morton order."""

    dim: int
    n_per_dir: int
    cell_ijk: np.ndarray  # [n_cells, dim] integer position of each active cell
    ijk_to_cell: np.ndarray  # dense lookup
    vertices: np.ndarray  # [n_cells, 2^dim, dim]
    n_dir: tuple = None  # cells per direction (subdivided_hyper_rectangle); None: n_per_dir in every direction

    @property
    def dirs(self):
        return self.n_dir if self.n_dir is not None else (self.n_per_dir,) * self.dim

    @property
    def n_cells(self):
        return self.cell_ijk.shape[0]

    def neighbor(self, cell: int, f: int) -> int:
        ax, side = f // 2, f % 2
        ijk = self.cell_ijk[cell].copy()
        ijk[ax] += 1 if side else -1
        if ijk[ax] < 0 or ijk[ax] >= self.dirs[ax]:
            return INVALID
        return int(self.ijk_to_cell[tuple(ijk)])

    @staticmethod
    def neighbor_of_neighbor(f: int, cell: int = -1) -> int:
        return f ^ 1

    @staticmethod
    def face_points_reversed(cell: int, f: int) -> bool:
        return False

    def distort(self, factor: float, seed: int = 0):
        """Random interior-vertex jitter, in the spirit of GridTools::distort_random as used by
        test/polydeal/exact_solutions_dgp.cc:306 (the RNG stream itself is not reproduced)."""
        rng = np.random.default_rng(seed)
        nd = self.dirs
        shape = tuple(m + 1 for m in nd) + (self.dim,)
        h = (self.vertices[0, -1, 0] - self.vertices[0, 0, 0])
        jitter = (rng.random(shape) * 2 - 1) * factor * h
        # keep boundary vertices fixed
        for c in range(self.dim):
            sl = [slice(None)] * (self.dim + 1)
            sl[c] = 0
            jitter[tuple(sl)] = 0
            sl[c] = nd[c]
            jitter[tuple(sl)] = 0
        for cell in range(self.n_cells):
            for v in range(2 ** self.dim):
                vid = tuple(int(self.cell_ijk[cell, c] + ((v >> c) & 1)) for c in range(self.dim))
                self.vertices[cell, v] += jitter[vid]
        return self


def _build_grid(dim, n, lo, hi, order):
    """n: cells per direction (int, or a tuple of dim ints for a rectangle - lexicographic order only)."""
    nd = tuple(int(m) for m in n) if isinstance(n, (tuple, list)) else (int(n),) * dim
    ijk = np.array(list(itertools.product(*[range(m) for m in nd[::-1]])), dtype=np.int64)[:, ::-1]  # x fastest
    if order == "morton":
        levels = int(round(math.log2(nd[0])))
        assert all(2 ** levels == m for m in nd)
        key = _morton_encode(ijk, dim, levels)
    else:
        key = np.zeros(len(ijk), dtype=np.int64)
        mul = 1
        for c in range(dim):
            key += ijk[:, c] * mul
            mul *= nd[c]
    perm = np.argsort(key, kind="stable")
    cell_ijk = ijk[perm]
    lut = np.zeros(nd, dtype=np.int64)
    lut[tuple(cell_ijk.T)] = np.arange(len(cell_ijk))
    lo = np.broadcast_to(np.asarray(lo, dtype=np.float64), (dim,))
    hi = np.broadcast_to(np.asarray(hi, dtype=np.float64), (dim,))
    h = (hi - lo) / np.asarray(nd, dtype=np.float64)
    verts = np.zeros((len(cell_ijk), 2 ** dim, dim))
    for v in range(2 ** dim):
        off = np.array([(v >> c) & 1 for c in range(dim)])
        verts[:, v, :] = lo + (cell_ijk + off) * h
    return Grid(dim, nd[0], cell_ijk, lut, verts, None if len(set(nd)) == 1 else nd)


class UnstructuredGrid:
    """2-D quadrilateral mesh with explicit adjacency: GridIn<2>::read_msh + Triangulation::refine_global  [deal.II], as the
    reference's callers use it on meshes/t3.msh (examples/minimal_SIP.cc:94-118, test/polydeal/poisson_sanity_check_03.cc:
    103-113).  Same interface as Grid where the handler needs it (neighbor, neighbor_of_neighbor, vertices, n_cells).
    vertex_id numbers the vertices globally: two cells that share an edge agree on its direction through it
    (face_points_reversed), which is what makes their face quadrature points coincide index by index, as deal.II's face
    orientation does (asserted by the reference: test/polydeal/reinit_cell_face_quad_pts.cc)."""

    FACE_VERTICES = ((0, 2), (1, 3), (0, 1), (2, 3))  # face f runs from the first to the second (free coordinate increasing)

    def __init__(self, vertices, vertex_id):
        self.dim = 2
        self.vertices = np.asarray(vertices, dtype=np.float64)  # [n_cells, 4, 2]
        self.vertex_id = np.asarray(vertex_id, dtype=np.int64)  # [n_cells, 4]
        self._adjacency()

    @property
    def n_cells(self):
        return self.vertices.shape[0]

    def _adjacency(self):
        n = self.n_cells
        self.nbr = np.full((n, 4), INVALID, dtype=np.int64)
        self.nofn = np.full((n, 4), INVALID, dtype=np.int64)
        first = {}
        for c in range(n):
            for f, (a, b) in enumerate(self.FACE_VERTICES):
                key = tuple(sorted((int(self.vertex_id[c, a]), int(self.vertex_id[c, b]))))
                if key in first:
                    oc, of = first[key]
                    assert self.nbr[oc, of] == INVALID, "an edge is shared by more than two cells"
                    self.nbr[c, f], self.nofn[c, f] = oc, of
                    self.nbr[oc, of], self.nofn[oc, of] = c, f
                else:
                    first[key] = (c, f)

    def neighbor(self, cell: int, f: int) -> int:
        return int(self.nbr[cell, f])

    def neighbor_of_neighbor(self, f: int, cell: int = -1) -> int:
        return int(self.nofn[cell, f])

    def face_points_reversed(self, cell: int, f: int) -> bool:
        a, b = self.FACE_VERTICES[f]
        return bool(self.vertex_id[cell, a] > self.vertex_id[cell, b])

    def refine_global(self, times: int = 1):
        """Children of cell i become cells 4i .. 4i+3, child c at ((c&1),(c>>1)) of the parent; new vertices at the edge
        midpoints and at the mean of the four vertices (FlatManifold on straight-sided cells)  [deal.II]."""
        for _ in range(times):
            nv = int(self.vertex_id.max()) + 1
            mids = {}
            new_xy = []

            def mid(c, a, b):
                nonlocal nv
                key = tuple(sorted((int(self.vertex_id[c, a]), int(self.vertex_id[c, b]))))
                if key not in mids:
                    mids[key] = (nv + len(new_xy), 0.5 * (self.vertices[c, a] + self.vertices[c, b]))
                    new_xy.append(mids[key][1])
                return mids[key]

            verts, ids = [], []
            for c in range(self.n_cells):
                V = [(int(self.vertex_id[c, v]), self.vertices[c, v]) for v in range(4)]
                EL, ER, EB, ET = mid(c, 0, 2), mid(c, 1, 3), mid(c, 0, 1), mid(c, 2, 3)
                C = (nv + len(new_xy), 0.25 * self.vertices[c].sum(axis=0))
                new_xy.append(C[1])
                for ch in ((V[0], EB, EL, C), (EB, V[1], C, ER), (EL, C, V[2], ET), (C, ER, ET, V[3])):
                    ids.append([k[0] for k in ch])
                    verts.append([k[1] for k in ch])
            self.vertices = np.asarray(verts, dtype=np.float64)
            self.vertex_id = np.asarray(ids, dtype=np.int64)
            self._adjacency()
        return self


def read_msh(path) -> UnstructuredGrid:
    """gmsh 4.1 ASCII, 4-node quadrilaterals (element type 3).  gmsh lists a quadrilateral counter-clockwise; deal.II's vertex
    order is lexicographic, i.e. nodes (0, 1, 3, 2)  [deal.II GridIn::read_msh]."""
    tok = open(path).read().split("\n")
    pos = {name: i for i, name in enumerate(tok) if name.startswith("$")}
    ver = float(tok[pos["$MeshFormat"] + 1].split()[0])
    assert 4.0 <= ver < 5.0, "only the ASCII format 4.x is read"
    nodes = {}
    i = pos["$Nodes"] + 1
    nblocks = int(tok[i].split()[0])
    i += 1
    for _ in range(nblocks):
        edim, _tag, parametric, nb = (int(t) for t in tok[i].split())
        tags = [int(tok[i + 1 + k]) for k in range(nb)]
        for k in range(nb):
            nodes[tags[k]] = [float(t) for t in tok[i + 1 + nb + k].split()[:3]]
        i += 1 + 2 * nb
    quads = []
    i = pos["$Elements"] + 1
    nblocks = int(tok[i].split()[0])
    i += 1
    for _ in range(nblocks):
        _edim, _tag, etype, nb = (int(t) for t in tok[i].split())
        for k in range(nb):
            row = [int(t) for t in tok[i + 1 + k].split()]
            if etype == 3:
                quads.append(row[1:5])
            else:
                assert etype in (15, 1), "quadrilateral meshes only"
        i += 1 + nb
    used = sorted({t for q in quads for t in q})
    vid = {t: k for k, t in enumerate(used)}
    verts, ids = [], []
    for q in quads:
        xy = np.array([nodes[t][:2] for t in q])
        area2 = sum(xy[v, 0] * xy[(v + 1) % 4, 1] - xy[v, 1] * xy[(v + 1) % 4, 0] for v in range(4))
        order = (0, 1, 3, 2) if area2 > 0 else (1, 0, 2, 3)  # a clockwise list is mirrored
        ids.append([vid[q[o]] for o in order])
        verts.append([nodes[q[o]][:2] for o in order])
    return UnstructuredGrid(verts, ids)


def hyper_cube_refined(dim, lo, hi, n_refine) -> Grid:
    """GridGenerator::hyper_cube(tria, lo, hi); tria.refine_global(n_refine)  [deal.II]."""
    return _build_grid(dim, 2 ** n_refine, lo, hi, "morton")


def subdivided_hyper_cube(dim, n, lo=0.0, hi=1.0) -> Grid:
    """GridGenerator::subdivided_hyper_cube(tria, n, lo, hi): lexicographic cells [deal.II]."""
    return _build_grid(dim, n, lo, hi, "lex")


def subdivided_hyper_rectangle(dim, repetitions, lo, hi) -> Grid:
    """GridGenerator::subdivided_hyper_rectangle(tria, repetitions, p1, p2): lexicographic cells [deal.II]."""
    return _build_grid(dim, tuple(repetitions), lo, hi, "lex")


# ---------------------------------------------------------------------------
# Q1 mapping of cells and faces (what FEValues/FEFaceValues on FE_Nothing provide in
# source/agglomeration_handler.cc:639-653 and :1146-1165)
# ---------------------------------------------------------------------------
def _q1_shape(dim, xi):
    """N[v](xi), dN[v, c](xi) for the multilinear map, vertex v at bits (v>>c)&1."""
    nv = 2 ** dim
    N = np.ones((xi.shape[0], nv))
    dN = np.ones((xi.shape[0], nv, dim))
    for v in range(nv):
        for c in range(dim):
            b = (v >> c) & 1
            f = xi[:, c] if b else 1.0 - xi[:, c]
            df = 1.0 if b else -1.0
            N[:, v] *= f
            for g in range(dim):
                dN[:, v, g] *= df if g == c else f
    return N, dN


def cell_quadrature(grid: Grid, cell: int, nq: int):
    """Real q-points and JxW of QGauss<dim>(nq) on a cell (x fastest)  [deal.II]."""
    dim = grid.dim
    x1, w1 = qgauss_1d(nq)
    pts = np.array(list(itertools.product(range(nq), repeat=dim)), dtype=np.int64)[:, ::-1]
    xi = x1[pts]
    w = np.prod(w1[pts], axis=1)
    N, dN = _q1_shape(dim, xi)
    X = grid.vertices[cell]
    x = N @ X
    J = np.einsum("qvc,vr->qrc", dN, X)  # J[q, real r, unit c]
    det = np.linalg.det(J)
    return x, w * np.abs(det)


def _face_unit_points(dim, f, nqf):
    """QProjector::project_to_face ordering  [deal.II]: 2-D faces run along the free axis;
    3-D: faces 0/1 -> (0|1, s, t), 2/3 -> (t, 0|1, s), 4/5 -> (s, t, 0|1), s fastest."""
    x1, w1 = qgauss_1d(nqf)
    ax, side = f // 2, f % 2
    if dim == 2:
        xi = np.zeros((nqf, 2))
        free = 1 - ax
        xi[:, free] = x1
        xi[:, ax] = side
        return xi, w1.copy(), [free]
    st = np.array(list(itertools.product(range(nqf), repeat=2)), dtype=np.int64)[:, ::-1]  # s fastest
    s, t = x1[st[:, 0]], x1[st[:, 1]]
    w = w1[st[:, 0]] * w1[st[:, 1]]
    xi = np.zeros((nqf * nqf, 3))
    if ax == 0:
        xi[:, 1], xi[:, 2] = s, t
        tang = [1, 2]
    elif ax == 1:
        xi[:, 2], xi[:, 0] = s, t
        tang = [2, 0]
    else:
        xi[:, 0], xi[:, 1] = s, t
        tang = [0, 1]
    xi[:, ax] = side
    return xi, w, tang


def face_quadrature(grid: Grid, cell: int, f: int, nqf: int):
    """Real q-points, JxW and outward unit normals of QGauss<dim-1>(nqf) on face f of a cell."""
    dim = grid.dim
    xi, w, tang = _face_unit_points(dim, f, nqf)
    if grid.face_points_reversed(cell, f):  # the direction both cells sharing the face agree on (UnstructuredGrid)
        xi, w = xi[::-1].copy(), w[::-1].copy()
    N, dN = _q1_shape(dim, xi)
    X = grid.vertices[cell]
    x = N @ X
    J = np.einsum("qvc,vr->qrc", dN, X)
    ax, side = f // 2, f % 2
    if dim == 2:
        t = J[:, :, tang[0]]
        area = np.linalg.norm(t, axis=1)
        nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)
    else:
        cr = np.cross(J[:, :, tang[0]], J[:, :, tang[1]])
        area = np.linalg.norm(cr, axis=1)
        nrm = cr
    nrm = nrm / np.linalg.norm(nrm, axis=1)[:, None]
    # orient outward: compare with the direction of increasing unit coordinate `ax`
    out_dir = J[:, :, ax] * (1.0 if side else -1.0)
    sign = np.sign(np.einsum("qr,qr->q", nrm, out_dir))
    nrm = nrm * sign[:, None]
    return x, w * area, nrm


# ---------------------------------------------------------------------------
# AgglomerationHandler restatement
# ---------------------------------------------------------------------------
class AgglomerationHandler:
    """Restates include/agglomeration_handler.h + source/agglomeration_handler.cc (serial paths).

    Polytope index = order of define_agglomerate calls (source/agglomeration_handler.cc:96);
    master = cells[0] (:56-59); get_agglomerate = slaves in insertion order then master
    (include/agglomeration_handler.h:1022-1032)."""

    def __init__(self, grid: Grid):
        self.grid = grid
        self.master_of = np.full(grid.n_cells, INVALID, dtype=np.int64)  # master cell index of each cell
        self.master_cells: list[int] = []  # master_cells_container, in polytope order
        self.master2polygon: dict[int, int] = {}
        self.master2slaves: dict[int, list[int]] = {}
        self.bboxes: list[tuple[np.ndarray, np.ndarray]] = []
        self.fe = None

    # -- source/agglomeration_handler.cc:44-104 + create_bounding_box :476-491
    def define_agglomerate(self, cells):
        cells = [int(c) for c in cells]
        assert len(cells) > 0
        master = cells[0]
        poly = len(self.master_cells)
        self.master_cells.append(master)
        self.master_of[master] = master
        self.master2slaves[master] = cells[1:]
        for c in cells[1:]:
            self.master_of[c] = master
        self.master2polygon[master] = poly
        V = self.grid.vertices[cells].reshape(-1, self.grid.dim)
        self.bboxes.append((V.min(axis=0), V.max(axis=0)))
        return poly

    @property
    def n_agglomerates(self):
        return len(self.master_cells)

    def get_agglomerate(self, poly: int):
        m = self.master_cells[poly]
        return self.master2slaves[m] + [m]  # slaves then master

    def polytope_of_cell(self, cell: int) -> int:
        return self.master2polygon[int(self.master_of[cell])]

    # -- source/agglomeration_handler.cc:326-379, 711-725 (+ SURVEY A2 numbering)
    def distribute_agglomerated_dofs(self, fe):
        assert np.all(self.master_of >= 0), "every cell must belong to an agglomerate"
        self.fe = fe
        n = fe.n_dofs_per_cell
        order = np.argsort(np.array(self.master_cells))  # masters in ascending cell index [deal.II]
        self.dof_offset = np.zeros(self.n_agglomerates, dtype=np.int64)
        for rank, poly in enumerate(order):
            self.dof_offset[poly] = rank * n
        self.n_dofs = n * self.n_agglomerates
        self._setup_connectivity()

    def dof_indices(self, poly: int):
        n = self.fe.n_dofs_per_cell
        return np.arange(self.dof_offset[poly], self.dof_offset[poly] + n)

    # -- source/agglomeration_handler.cc:495-527 and :1253-1645 (serial branches)
    def _setup_connectivity(self):
        g = self.grid
        nP = self.n_agglomerates
        self.n_faces = [0] * nP
        self.face_info: dict[tuple[int, int], tuple[bool, int]] = {}  # (P,f) -> (at_boundary, neighbour polytope)
        self.interface: dict[tuple[int, int], list[tuple[int, int]]] = {}
        visited = set()  # GLOBAL (cell, face)  include/agglomeration_handler.h:117-118
        nfaces_cell = 2 * g.dim
        for P, master in enumerate(self.master_cells):
            seen = set()
            for cell in self.get_agglomerate(P):
                for f in range(nfaces_cell):
                    nb = g.neighbor(cell, f)
                    if nb != INVALID:
                        if self.master_of[nb] != self.master_of[cell]:
                            Q = self.polytope_of_cell(nb)
                            nof = g.neighbor_of_neighbor(f, cell)
                            if Q not in seen:
                                self.face_info[(P, self.n_faces[P])] = (False, Q)
                                self.n_faces[P] += 1
                                seen.add(Q)
                            if (cell, f) not in visited:
                                self.interface.setdefault((P, Q), []).append((cell, f))
                                visited.add((cell, f))
                            if (nb, nof) not in visited:
                                self.interface.setdefault((Q, P), []).append((nb, nof))
                                visited.add((nb, nof))
                    else:
                        if BDRY_KEY not in seen:
                            self.face_info[(P, self.n_faces[P])] = (True, INVALID)
                            self.n_faces[P] += 1
                            seen.add(BDRY_KEY)
                        if (cell, f) not in visited:
                            self.interface.setdefault((P, P), []).append((cell, f))
                            visited.add((cell, f))

    # -- include/agglomeration_accessor.h
    def at_boundary(self, P, f):  # :736-772
        return self.face_info[(P, f)][0]

    def neighbor(self, P, f):  # :335-422
        return self.face_info[(P, f)][1]

    def neighbor_of_agglomerated_neighbor(self, P, f):  # :426-481 (linear scan)
        if self.at_boundary(P, f):
            return INVALID  # numbers::invalid_unsigned_int (:476-480)
        Q = self.neighbor(P, f)
        for fo in range(self.n_faces[Q]):
            if not self.at_boundary(Q, fo) and self.neighbor(Q, fo) == P:
                return fo
        return INVALID

    def diameter(self, P):  # :582-601
        lo, hi = self.bboxes[P]
        return float(np.linalg.norm(hi - lo))

    def volume(self, P):  # :617-632
        lo, hi = self.bboxes[P]
        return float(np.prod(hi - lo))

    def master_index(self, P):
        return self.master_cells[P]

    # -- handler protocol shared with the product mirror (tests/golden_cases.py)
    def n_faces_of(self, P):  # include/agglomeration_accessor.h:324-331
        return self.n_faces[P]

    def interface_list(self, P, Q):  # polytope_cache.interface (agglomeration_handler.h:355-356)
        return self.interface[(P, Q)]

    def bbox(self, P):
        return self.bboxes[P]

    def master_slave_value(self, cell):  # master_slave_relationships (agglomeration_handler.h:688)
        m = int(self.master_of[cell])
        return -1 if m == cell else m

    def sparsity_rows(self):
        n = self.fe.n_dofs_per_cell
        blocks = self.sparsity_blocks()
        rows = [None] * self.n_dofs
        for P in range(self.n_agglomerates):
            offs = sorted(self.dof_offset[Q] for Q in blocks[P])
            cols = np.concatenate([np.arange(o, o + n) for o in offs])
            for i in range(n):
                rows[self.dof_offset[P] + i] = cols
        return rows

    def volume_jxw_sum(self, P):
        return float(np.sum(self.agglomerated_quadrature(P)[1]))

    def face_jxw_sum(self, P, f):
        return float(np.sum(self.reinit_face(P, f)["JxW"]))

    # -- source/agglomeration_handler.cc:210-236
    def initialize_fe_values(self, nq: int, nqf: int):
        self.nq, self.nqf = nq, nqf

    # -- source/agglomeration_handler.cc:622-707
    def agglomerated_quadrature(self, P):
        xs, ws = [], []
        for cell in self.get_agglomerate(P):
            x, w = cell_quadrature(self.grid, cell, self.nq)
            xs.append(x)
            ws.append(w)
        x = np.concatenate(xs)
        w = np.concatenate(ws)
        return x, w

    def real_to_unit(self, P, x):  # BoundingBox::real_to_unit, :698-704
        lo, hi = self.bboxes[P]
        return (x - lo) / (hi - lo)

    # -- reinit(polytope): source/agglomeration_handler.cc:729-767 + source/mapping_box.cc:393-439, 507-531
    def reinit(self, P):
        x, w = self.agglomerated_quadrature(P)
        lo, hi = self.bboxes[P]
        val, ugrad = self.fe.shape(self.real_to_unit(P, x))
        grad = ugrad * (1.0 / (hi - lo))  # inverse_cell_extents, mapping_box.cc:210-222, 528-530
        return dict(x=x, JxW=w, val=val, grad=grad)

    # -- reinit_master: source/agglomeration_handler.cc:1103-1243 + mapping_box.cc:465-503
    def reinit_face(self, P, f):
        at_b, Q = self.face_info[(P, f)]
        key = (P, P) if at_b else (P, Q)
        xs, ws, ns = [], [], []
        for cell, lf in self.interface[key]:
            x, w, nrm = face_quadrature(self.grid, cell, lf, self.nqf)
            xs.append(x)
            ws.append(w)
            ns.append(nrm)
        x, w, nrm = np.concatenate(xs), np.concatenate(ws), np.concatenate(ns)
        lo, hi = self.bboxes[P]
        val, ugrad = self.fe.shape(self.real_to_unit(P, x))
        grad = ugrad * (1.0 / (hi - lo))
        return dict(x=x, JxW=w, normal=nrm, val=val, grad=grad)

    def reinit_interface(self, P, Q, f, nofn):  # :805-834
        return self.reinit_face(P, f), self.reinit_face(Q, nofn)

    # -- create_agglomeration_sparsity_pattern: source/agglomeration_handler.cc:910-1022
    def sparsity_blocks(self):
        """For each polytope: sorted list of coupled polytopes (own + valid neighbours)."""
        out = []
        for P in range(self.n_agglomerates):
            s = {P}
            for f in range(self.n_faces[P]):
                if not self.at_boundary(P, f):
                    s.add(self.neighbor(P, f))
            out.append(s)
        return out

    def sparsity_pattern(self, diag_first: bool = True):
        """CSR (rowptr, colind) of the DG pattern; serial deal.II SparsityPattern stores the
        diagonal first, then ascending columns  [deal.II]."""
        n = self.fe.n_dofs_per_cell
        blocks = self.sparsity_blocks()
        rowptr = np.zeros(self.n_dofs + 1, dtype=np.int64)
        rows_cols = [None] * self.n_dofs
        for P in range(self.n_agglomerates):
            offs = sorted(self.dof_offset[Q] for Q in blocks[P])
            cols = np.concatenate([np.arange(o, o + n) for o in offs])
            for i in range(n):
                r = self.dof_offset[P] + i
                if diag_first:
                    c = np.concatenate([[r], cols[cols != r]])
                else:
                    c = cols
                rows_cols[r] = c
                rowptr[r + 1] = len(c)
        rowptr = np.cumsum(rowptr)
        colind = np.concatenate(rows_cols).astype(np.int64)
        return rowptr, colind


# ---------------------------------------------------------------------------
# SIP assembly (the hot path), in the reference's loop structure
# ---------------------------------------------------------------------------
@dataclass
class SipVariant:
    """Scalars that differ between the reference's callers (SURVEY.md 8(a) variants table)."""

    name: str = "assemble_dg_matrix"
    penalty_constant: float | None = None  # None -> 10 (p+dim)(p+1)  (include/poly_utils.h:2018-2019)
    owner_rule: str = "id"  # 'id' (poly_utils.h:2089) or 'index' (examples/poisson.cc:841)
    h_rule: str = "diameter_in"  # 'diameter_in' | 'one' | 'max_inv' (examples/minimal_SIP.cc:256-259)
    boundary: str = "nitsche"  # 'nitsche' | 'zero' (examples/minimal_SIP.cc:230-248)
    reaction_c: float = 0.0  # examples/diffusion_reaction.cc:495-501


def variant_assemble_dg_matrix():
    return SipVariant()


def variant_poisson_example(fe):
    # examples/poisson.cc:476: penalty_constant = 10 (p+1)(p+dim); owner index()<index() (:841)
    return SipVariant("poisson.cc", 10.0 * (fe.degree + 1) * (fe.degree + fe.dim), "index", "diameter_in")


def variant_minimal_sip_test():
    # test/polydeal/minimal_SIP_Poisson.cc:101, 308: penalty 20, hf = 1, index()<index()
    return SipVariant("minimal_SIP_Poisson", 20.0, "index", "one")


def variant_minimal_sip_example():
    # examples/minimal_SIP.cc:256-262: sigma = 10*max(1/h_in,1/h_out), boundary zeroed
    return SipVariant("minimal_SIP.cc", 10.0, "index", "max_inv", "zero")


def variant_diffusion_reaction(fe):
    # examples/diffusion_reaction.cc:366, 515: 10 p^2 / h_in; id()<id(); reaction c = 0.5
    return SipVariant("diffusion_reaction.cc", 10.0 * fe.degree ** 2, "id", "diameter_in", "nitsche", 0.5)


def _penalty_constant(ah, var):
    if var.penalty_constant is not None:
        return var.penalty_constant
    fe = ah.fe
    return 10.0 * (fe.degree + fe.dim) * (fe.degree + 1)


def _owns(ah, var, P, Q):
    if var.owner_rule == "index":
        return P < Q
    # CellId order of the masters: on a single-coarse-cell refined mesh CellId order == Morton
    # active index order; for lexicographic grids we use the master cell index as the id.
    return ah.master_cells[P] < ah.master_cells[Q]


def face_sigma(ah, var, P, Q=None):
    C = _penalty_constant(ah, var)
    if var.h_rule == "one":
        return C
    if var.h_rule == "max_inv" and Q is not None:
        return C * max(1.0 / ah.diameter(P), 1.0 / ah.diameter(Q))
    return C / abs(ah.diameter(P))


def assemble_blocks(ah: AgglomerationHandler, var: SipVariant):
    """Runs the reference's polytope/face loops (include/poly_utils.h:2034-2193,
    examples/poisson.cc:733-987) and returns the dense blocks keyed by (P_row, P_col)."""
    n = ah.fe.n_dofs_per_cell
    blocks: dict[tuple[int, int], np.ndarray] = {}

    def add(P, Q, M):
        if (P, Q) in blocks:
            blocks[(P, Q)] += M
        else:
            blocks[(P, Q)] = M.copy()

    for P in range(ah.n_agglomerates):
        fv = ah.reinit(P)
        # volume: poly_utils.h:2040-2052
        cell = np.einsum("qic,qjc,q->ij", fv["grad"], fv["grad"], fv["JxW"])
        if var.reaction_c != 0.0:
            cell += var.reaction_c * np.einsum("qi,qj,q->ij", fv["val"], fv["val"], fv["JxW"])
        for f in range(ah.n_faces[P]):
            if ah.at_boundary(P, f):
                if var.boundary == "zero":
                    continue
                ff = ah.reinit_face(P, f)
                sig = face_sigma(ah, var, P)
                g = np.einsum("qic,qc->qi", ff["grad"], ff["normal"])
                v, w = ff["val"], ff["JxW"]
                # poly_utils.h:2065-2084
                cell += np.einsum("qi,qj,q->ij", -v, g, w) + np.einsum("qi,qj,q->ij", -g, v, w) \
                    + sig * np.einsum("qi,qj,q->ij", v, v, w)
            else:
                Q = ah.neighbor(P, f)
                if _owns(ah, var, P, Q):
                    nofn = ah.neighbor_of_agglomerated_neighbor(P, f)
                    assert ah.neighbor(Q, nofn) == P
                    f0, f1 = ah.reinit_interface(P, Q, f, nofn)
                    sig = face_sigma(ah, var, P, Q)
                    nrm = f0["normal"]  # normal of side 0 for all four blocks (poly_utils.h:1881)
                    g0 = np.einsum("qic,qc->qi", f0["grad"], nrm)
                    g1 = np.einsum("qic,qc->qi", f1["grad"], nrm)
                    v0, v1 = f0["val"], f1["val"]
                    w0, w1 = f0["JxW"], f1["JxW"]
                    e = lambda a, b, w: np.einsum("qi,qj,q->ij", a, b, w)
                    # poly_utils.h:1891-1922
                    M11 = -0.5 * e(g0, v0, w0) - 0.5 * e(v0, g0, w0) + sig * e(v0, v0, w0)
                    M12 = 0.5 * e(g0, v1, w1) - 0.5 * e(v0, g1, w1) - sig * e(v0, v1, w1)
                    M21 = -0.5 * e(g1, v0, w1) + 0.5 * e(v1, g0, w1) - sig * e(v1, v0, w1)
                    M22 = 0.5 * e(g1, v1, w1) + 0.5 * e(v1, g1, w1) + sig * e(v1, v1, w1)
                    add(P, P, M11)
                    add(P, Q, M12)
                    add(Q, P, M21)
                    add(Q, Q, M22)
        add(P, P, cell)
    return blocks


def assemble_csr(ah: AgglomerationHandler, var: SipVariant, diag_first: bool = True):
    """Scatter the blocks into the deal.II-layout CSR (A11: distribute_local_to_global)."""
    rowptr, colind = ah.sparsity_pattern(diag_first)
    values = np.zeros(len(colind))
    n = ah.fe.n_dofs_per_cell
    blocks = assemble_blocks(ah, var)
    for (P, Q), M in blocks.items():
        for i in range(n):
            r = ah.dof_offset[P] + i
            cols = colind[rowptr[r]:rowptr[r + 1]]
            for j in range(n):
                c = ah.dof_offset[Q] + j
                if diag_first and c == r:
                    pos = 0
                else:
                    lo = 1 if diag_first else 0
                    pos = lo + int(np.searchsorted(cols[lo:], c))
                assert cols[pos] == c
                values[rowptr[r] + pos] += M[i, j]
    return rowptr, colind, values


def assemble_rhs(ah: AgglomerationHandler, var: SipVariant, f=None, g=None):
    """Right-hand side as the callers assemble it next to the matrix (examples/poisson.cc:745-759, 788-828):
    volume sum_q phi_i f JxW and Nitsche boundary sum_q (sigma g phi_i - grad phi_i . n g) JxW."""
    b = np.zeros(ah.n_dofs)
    for P in range(ah.n_agglomerates):
        idx = ah.dof_indices(P)
        if f is not None:
            fv = ah.reinit(P)
            b[idx] += np.einsum("qi,q->i", fv["val"], f(fv["x"]) * fv["JxW"])
        if g is not None and var.boundary != "zero":
            for fc in range(ah.n_faces[P]):
                if ah.at_boundary(P, fc):
                    ff = ah.reinit_face(P, fc)
                    sig = face_sigma(ah, var, P)
                    gn = np.einsum("qic,qc->qi", ff["grad"], ff["normal"])
                    b[idx] += np.einsum("qi,q->i", sig * ff["val"] - gn, g(ff["x"]) * ff["JxW"])
    return b


def csr_to_dense(rowptr, colind, values, n):
    A = np.zeros((n, n))
    for r in range(n):
        A[r, colind[rowptr[r]:rowptr[r + 1]]] += values[rowptr[r]:rowptr[r + 1]]
    return A


def assemble_dense(ah, var):
    n = ah.fe.n_dofs_per_cell
    A = np.zeros((ah.n_dofs, ah.n_dofs))
    for (P, Q), M in assemble_blocks(ah, var).items():
        A[ah.dof_offset[P]:ah.dof_offset[P] + n, ah.dof_offset[Q]:ah.dof_offset[Q] + n] += M
    return A


# ---------------------------------------------------------------------------
# Convenience: structured block agglomeration (stand-in for METIS / R-tree levels)
# ---------------------------------------------------------------------------
def block_agglomerates(grid: Grid, b: int):
    """Lists of cells forming b^dim blocks, in mesh order inside a block (so the master is the
    lowest index, as collect_cells_for_agglomeration yields: include/poly_utils.h:532-538);
    blocks enumerated lexicographically (x fastest)."""
    dim = grid.dim
    assert all(m % b == 0 for m in grid.dirs)
    nbd = [m // b for m in grid.dirs]
    out = []
    for bijk in itertools.product(*[range(m) for m in nbd[::-1]]):
        bijk = bijk[::-1]
        cells = []
        for off in itertools.product(range(b), repeat=dim):
            off = off[::-1]
            ijk = tuple(bijk[c] * b + off[c] for c in range(dim))
            cells.append(int(grid.ijk_to_cell[ijk]))
        out.append(sorted(cells))
    return out


def interpolate_nodal(ah: AgglomerationHandler, func):
    """Coefficient vector of the FE_DGQ nodal interpolant of `func` on every polytope's bbox."""
    fe = ah.fe
    assert isinstance(fe, FE_DGQ)
    u = np.zeros(ah.n_dofs)
    nodes = fe.nodes[fe.multi_index]  # [n, dim]
    for P in range(ah.n_agglomerates):
        lo, hi = ah.bboxes[P]
        x = lo + nodes * (hi - lo)
        u[ah.dof_indices(P)] = func(x)
    return u


# ===================================================================================================
# Post-processing
# ===================================================================================================
def evaluate_at(ah: AgglomerationHandler, u, P: int, x):
    """u_h and grad u_h of polytope P at real points x [N, dim]: basis on the bounding box, values through
    BoundingBox::real_to_unit, gradients scaled by the inverse box extents.
    include/poly_utils.h:1196-1233 (interpolate_to_fine_grid) and :1699-1711 (compute_global_error)."""
    lo, hi = ah.bboxes[P]
    val, ugrad = ah.fe.shape(ah.real_to_unit(P, np.atleast_2d(x)))
    coef = np.asarray(u)[ah.dof_indices(P)]
    return val @ coef, np.einsum("qic,i->qc", ugrad * (1.0 / (hi - lo)), coef)


def compute_global_error(ah: AgglomerationHandler, u, exact, exact_grad=None):
    """L2 error and H1-seminorm error over the polytopes' own quadrature:
    include/poly_utils.h:1647-1750 (sum_q (u-u_h)^2 JxW, sum_q |grad u - grad u_h|^2 JxW, sqrt of the sums)."""
    l2 = h1 = 0.0
    for P in range(ah.n_agglomerates):
        fv = ah.reinit(P)
        coef = np.asarray(u)[ah.dof_indices(P)]
        uh = fv["val"] @ coef
        l2 += float(np.sum((exact(fv["x"]) - uh) ** 2 * fv["JxW"]))
        if exact_grad is not None:
            gh = np.einsum("qic,i->qc", fv["grad"], coef)
            h1 += float(np.sum(np.sum((exact_grad(fv["x"]) - gh) ** 2, axis=1) * fv["JxW"]))
    return np.sqrt(l2), (np.sqrt(h1) if exact_grad is not None else None)


def fill_injection_matrix(coarse_ah: AgglomerationHandler, fine_ah: AgglomerationHandler):
    """Dense injection matrix [n_fine_dofs, n_coarse_dofs] from the coarse polytopal space into the fine one:
    include/utils.h:95-270.  Block (child F, parent C): local(i, j) = phi^C_j(coarse_bbox.real_to_unit(
    fine_bbox.unit_to_real(unit support point i)))  (:212-229).  Children = fine polytopes nested in C
    (the reference reads them from the R-tree hierarchy, :169, 203)."""
    fe = coarse_ah.fe
    assert isinstance(fe, FE_DGQ)
    nodes = fe.nodes[fe.multi_index]  # unit support points [n, dim]
    M = np.zeros((fine_ah.n_dofs, coarse_ah.n_dofs))
    for F in range(fine_ah.n_agglomerates):
        cells = fine_ah.get_agglomerate(F)
        C = coarse_ah.polytope_of_cell(cells[0])
        assert all(coarse_ah.polytope_of_cell(c) == C for c in cells), "not nested"
        lo, hi = fine_ah.bboxes[F]
        real = lo + nodes * (hi - lo)  # BoundingBox::unit_to_real
        val, _ = fe.shape(coarse_ah.real_to_unit(C, real))
        M[np.ix_(fine_ah.dof_indices(F), coarse_ah.dof_indices(C))] = val
    return M
