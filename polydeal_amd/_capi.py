"""ctypes binding of the C ABI in include/polydeal_hip.h (libpolydeal_hip.so, built in-tree).

There is no CPU fallback: if the shared library is missing or no HIP device is present every compute
entry point raises.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDH_LIB: another build of the library (diagnostic builds with experiment switches; tools/README.md)
LIB_PATH = os.environ.get("PDH_LIB") or os.path.join(_HERE, "lib", "libpolydeal_hip.so")

PDH_BASIS_DGQ = 0
PDH_BASIS_AGGLODGP = 1
PDH_OK = 0
PDH_EINVAL, PDH_EUNSUPPORTED, PDH_EDEVICE, PDH_ESTATE = -1, -2, -3, -4


class PdhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pdh error %d: %s" % (code, msg))
        self.code = code


class pdh_problem(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("degree", C.c_int32), ("basis", C.c_int32), ("n_agg", C.c_int32),
        ("n_faces", C.c_int32), ("n_rows", C.c_int32), ("diag_first", C.c_int32), ("local", C.c_int32),
        ("reaction_c", C.c_double),
        ("bbox", C.c_void_p), ("dof_offset", C.c_void_p),
        ("vq_ptr", C.c_void_p), ("vq_x", C.c_void_p), ("vq_w", C.c_void_p),
        ("face_in", C.c_void_p), ("face_out", C.c_void_p), ("fq_ptr", C.c_void_p),
        ("fq_x", C.c_void_p), ("fq_n", C.c_void_p), ("fq_w", C.c_void_p), ("fq_w_out", C.c_void_p),
        ("face_sigma", C.c_void_p),
        ("rowptr", C.c_void_p), ("colind", C.c_void_p),
        ("col_offset", C.c_void_p), ("agg_rank", C.c_void_p),
        ("vq_tensor_n", C.c_int32), ("fq_tensor_n", C.c_int32),
    ]


EXPORTS = [
    "pdh_create", "pdh_destroy", "pdh_last_error", "pdh_set_problem", "pdh_set_problem_local",
    "pdh_assemble_device", "pdh_assemble", "pdh_assemble_sip", "pdh_assemble_sip_local",
    "pdh_device_values", "pdh_synchronize", "pdh_stream", "pdh_set_profiling", "pdh_kernel_times_ms",
    "pdh_problem_stats", "pdh_check_problem", "pdh_version", "pdh_assemble_rhs", "pdh_kernel_work", "pdh_evaluate", "pdh_shape_values", "pdh_set_algorithm", "pdh_algorithm_in_use", "pdh_set_overlap",
    "pdh_set_exchange_mode", "pdh_exchange_layout", "pdh_exchange_get_send", "pdh_exchange_apply", "pdh_set_stream",
    "pdh_check_exchange", "pdh_copy_values", "pdh_check_rows", "pdh_values_checksum",
    "pdh_assemble_rhs_device", "pdh_evaluate_device", "pdh_shape_values_device",
    "pdh_global_error", "pdh_global_error_device", "pdh_rows_kernel_in_use", "pdh_check_terms", "pdh_terms_merge_stats", "pdh_set_problem_cartesian",
]

_lib = None


class _Tolerant:
    """Attribute access that ignores entry points an OLDER build of the library lacks (tools/ab_bench.py loads two builds)."""

    class _Sink:
        argtypes = restype = None

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        try:
            return getattr(self._lib, name)
        except AttributeError:
            return _Tolerant._Sink()


def _bind(lib):
    P = C.POINTER
    real, lib = lib, _Tolerant(lib)
    lib.pdh_create.argtypes = [P(C.c_void_p), C.c_int]
    lib.pdh_destroy.argtypes = [C.c_void_p]
    lib.pdh_destroy.restype = None
    lib.pdh_last_error.argtypes = [C.c_void_p]
    lib.pdh_last_error.restype = C.c_char_p
    lib.pdh_set_problem.argtypes = [C.c_void_p, P(pdh_problem)]
    lib.pdh_set_problem_local.argtypes = [C.c_void_p, P(pdh_problem), C.c_int32, C.c_int32]
    lib.pdh_set_problem_cartesian.argtypes = [C.c_void_p, P(pdh_problem), C.c_void_p, C.c_int32, C.c_int32]
    lib.pdh_assemble_device.argtypes = [C.c_void_p]
    lib.pdh_assemble.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdh_assemble_sip.argtypes = [C.c_void_p, P(pdh_problem), C.c_void_p]
    lib.pdh_assemble_sip_local.argtypes = [C.c_void_p, P(pdh_problem), C.c_int32, C.c_int32, C.c_void_p]
    lib.pdh_assemble_rhs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pdh_evaluate.argtypes = [C.c_void_p] * 6
    lib.pdh_set_algorithm.argtypes = [C.c_void_p, C.c_int]
    lib.pdh_set_overlap.argtypes = [C.c_void_p, C.c_int]
    lib.pdh_algorithm_in_use.argtypes = [C.c_void_p]
    lib.pdh_rows_kernel_in_use.argtypes = [C.c_void_p]
    lib.pdh_check_terms.argtypes = [P(pdh_problem), C.c_int32, C.c_int32, P(C.c_int64)]
    lib.pdh_terms_merge_stats.argtypes = [C.c_void_p, P(C.c_int64)]
    lib.pdh_set_exchange_mode.argtypes = [C.c_void_p, C.c_int]
    lib.pdh_exchange_layout.argtypes = [C.c_void_p, C.c_int, P(C.c_int64), P(C.c_int64)]
    lib.pdh_exchange_get_send.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdh_exchange_apply.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdh_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdh_check_exchange.argtypes = [P(pdh_problem), C.c_int32, C.c_int32, C.c_int, P(C.c_int64), P(C.c_int64)]
    lib.pdh_copy_values.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdh_check_rows.argtypes = [P(pdh_problem), C.c_int32, C.c_int32]
    lib.pdh_values_checksum.argtypes = [C.c_void_p, P(C.c_double)]
    lib.pdh_assemble_rhs_device.argtypes = [C.c_void_p] * 4
    lib.pdh_evaluate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.pdh_shape_values_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p]
    lib.pdh_shape_values.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    lib.pdh_global_error.argtypes = [C.c_void_p] * 7 + [P(C.c_double)]
    lib.pdh_global_error_device.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3 + [P(C.c_double)]
    lib.pdh_device_values.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_int64)]
    lib.pdh_synchronize.argtypes = [C.c_void_p]
    lib.pdh_stream.argtypes = [C.c_void_p]
    lib.pdh_stream.restype = C.c_void_p
    lib.pdh_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.pdh_kernel_times_ms.argtypes = [C.c_void_p, P(C.c_float), P(C.c_int)]
    lib.pdh_problem_stats.argtypes = [C.c_void_p, P(C.c_int64)]
    lib.pdh_kernel_work.argtypes = [C.c_void_p, P(C.c_int64)]
    lib.pdh_check_problem.argtypes = [P(pdh_problem), C.c_int32, C.c_int32, P(C.c_int64)]
    lib.pdh_version.restype = C.c_char_p
    return real


def load_library(path=None):
    """Loads libpolydeal_hip.so; raises if the HIP extension has not been built.  `path` loads another build
    of the library next to the default one (tools/ab_bench.py compares two builds in one process)."""
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "polydeal_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C polydeal_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


_DTYPES = {
    "bbox": np.float64, "dof_offset": np.int32, "vq_ptr": np.int64, "vq_x": np.float64, "vq_w": np.float64,
    "face_in": np.int32, "face_out": np.int32, "fq_ptr": np.int64, "fq_x": np.float64, "fq_n": np.float64,
    "fq_w": np.float64, "fq_w_out": np.float64, "face_sigma": np.float64, "rowptr": np.int64, "colind": np.int32,
    "col_offset": np.int32, "agg_rank": np.int32,
}


class Problem:
    """Owns the NumPy arrays behind a pdh_problem (keeps them alive while the struct is in use)."""

    def __init__(self, *, dim, degree, basis, n_agg, n_faces, n_rows, diag_first=1, reaction_c=0.0, local=0, vq_tensor_n=0,
                 fq_tensor_n=0, **arrays):
        self.arrays = {}
        self.c = pdh_problem()
        self.c.dim, self.c.degree, self.c.basis = dim, degree, basis
        self.c.n_agg, self.c.n_faces, self.c.n_rows = n_agg, n_faces, n_rows
        self.c.diag_first, self.c.reaction_c, self.c.local = int(diag_first), float(reaction_c), int(local)
        self.c.vq_tensor_n = int(vq_tensor_n)
        self.c.fq_tensor_n = int(fq_tensor_n)
        for name, dt in _DTYPES.items():
            a = arrays.get(name)
            if a is None:
                setattr(self.c, name, None)
                continue
            a = np.ascontiguousarray(a, dtype=dt)
            self.arrays[name] = a
            setattr(self.c, name, a.ctypes.data)

    @property
    def nnz(self):
        return int(self.arrays["rowptr"][-1])

    def check(self, row_begin=0, row_end=None):
        lib = load_library()
        stats = (C.c_int64 * 8)()
        rc = lib.pdh_check_problem(C.byref(self.c), row_begin, self.c.n_rows if row_end is None else row_end, stats)
        if rc != PDH_OK:
            raise PdhError(rc, lib.pdh_last_error(None).decode())
        return list(stats)


class Context:
    """pdh_ctx wrapper (one per device and host thread)."""

    def __init__(self, device=0, lib_path=None):
        self.lib = load_library(lib_path)
        h = C.c_void_p()
        rc = self.lib.pdh_create(C.byref(h), device)
        if rc != PDH_OK:
            raise PdhError(rc, self.lib.pdh_last_error(None).decode())
        self.h = h
        self.n_values = 0

    def _chk(self, rc):
        if rc != PDH_OK:
            raise PdhError(rc, self.lib.pdh_last_error(self.h).decode())

    def set_problem(self, prob, row_begin=0, row_end=None):
        """prob: a Problem (NumPy-backed) or anything with a `.c` pdh_problem (e.g. handler.FlatView)."""
        row_end = prob.c.n_rows if row_end is None else row_end
        cart = getattr(prob, "cartesian", None)
        if cart:  # a flatten_cartesian view: the points are generated on the device
            self._chk(self.lib.pdh_set_problem_cartesian(self.h, C.byref(prob.c), C.c_void_p(cart), row_begin, row_end))
        else:
            self._chk(self.lib.pdh_set_problem_local(self.h, C.byref(prob.c), row_begin, row_end))
        self.n_values = self.stats()["n_values"]
        off = np.array((C.c_int32 * prob.c.n_agg).from_address(int(prob.c.dof_offset)), dtype=np.int64)
        self._owned = (off >= row_begin) & (off < row_end)  # polytopes whose rows live here

    def owned_point_mask(self, pt_ptr):
        """Boolean mask over per-polytope CSR points: True for the points of polytopes owned by this context."""
        ptr = np.asarray(pt_ptr, dtype=np.int64)
        return np.repeat(self._owned, np.diff(ptr))

    def assemble_device(self):
        self._chk(self.lib.pdh_assemble_device(self.h))

    def synchronize(self):
        self._chk(self.lib.pdh_synchronize(self.h))

    def assemble(self):
        out = np.empty(self.n_values, dtype=np.float64)
        self._chk(self.lib.pdh_assemble(self.h, out.ctypes.data))
        return out

    def checksum(self):
        """{sum, abs_sum, max_abs, non_finite} of the values as they stand in HBM (one device pass, nothing copied back)."""
        o = (C.c_double * 4)()
        self._chk(self.lib.pdh_values_checksum(self.h, o))
        return {"sum": float(o[0]), "abs_sum": float(o[1]), "max_abs": float(o[2]), "non_finite": int(o[3])}

    def values(self):
        """The CSR values as they stand in HBM (no re-assembly)."""
        out = np.empty(self.n_values, dtype=np.float64)
        self._chk(self.lib.pdh_copy_values(self.h, out.ctypes.data))
        return out

    def assemble_rhs(self, f_vol=None, g_bdry=None):
        """rhs of the owned rows; f_vol / g_bdry sampled at the caller's volume / face quadrature points."""
        n_rows = self.stats()["n_owned_agg"] * self.stats()["dofs_per_cell"]
        out = np.empty(n_rows, dtype=np.float64)
        fv = None if f_vol is None else np.ascontiguousarray(f_vol, dtype=np.float64)
        gb = None if g_bdry is None else np.ascontiguousarray(g_bdry, dtype=np.float64)
        self._chk(self.lib.pdh_assemble_rhs(self.h, None if fv is None else fv.ctypes.data,
                                            None if gb is None else gb.ctypes.data, out.ctypes.data))
        return out

    def assemble_rhs_device(self, d_f_vol, d_g_bdry, d_rhs):
        """Device pointers (ints or None), caller order; asynchronous on the context's stream."""
        self._chk(self.lib.pdh_assemble_rhs_device(self.h, C.c_void_p(d_f_vol or 0), C.c_void_p(d_g_bdry or 0), C.c_void_p(d_rhs)))

    def evaluate_device(self, d_solution, d_pt_ptr, d_pts, n_points, d_u, d_grad=None):
        self._chk(self.lib.pdh_evaluate_device(self.h, C.c_void_p(d_solution), C.c_void_p(d_pt_ptr), C.c_void_p(d_pts), n_points,
                                               C.c_void_p(d_u), C.c_void_p(d_grad or 0)))

    def evaluate(self, solution, pt_ptr, pts, want_grad=False):
        """u_h (and grad u_h) at caller-given real points; pt_ptr [n_agg+1], pts [dim][N]."""
        sol = np.ascontiguousarray(solution, dtype=np.float64)
        ptr = np.ascontiguousarray(pt_ptr, dtype=np.int64)
        p = np.ascontiguousarray(pts, dtype=np.float64)
        n = int(ptr[-1])
        u = np.zeros(n)
        g = np.zeros((p.shape[0], n)) if want_grad else None
        self._chk(self.lib.pdh_evaluate(self.h, sol.ctypes.data, ptr.ctypes.data, p.ctypes.data, u.ctypes.data,
                                        None if g is None else g.ctypes.data))
        return (u, g) if want_grad else u

    def global_error_sums(self, solution, pt_ptr, pts, w, exact_u, exact_grad):
        """(sum w (u - u_h)^2, sum w |grad u - grad u_h|^2) over the owned polytopes, formed on the device
        (pdh_global_error); exact_u [N], exact_grad [dim][N] sampled at pts [dim][N]."""
        sol = np.ascontiguousarray(solution, dtype=np.float64)
        ptr = np.ascontiguousarray(pt_ptr, dtype=np.int64)
        p = np.ascontiguousarray(pts, dtype=np.float64)
        ww = np.ascontiguousarray(w, dtype=np.float64)
        eu = np.ascontiguousarray(exact_u, dtype=np.float64)
        eg = np.ascontiguousarray(exact_grad, dtype=np.float64)
        n = int(ptr[-1])
        if ww.shape != (n,) or eu.shape != (n,) or eg.shape != (p.shape[0], n) or p.shape[1] != n:
            raise ValueError("w / exact_u [N], exact_grad / pts [dim][N] with N = pt_ptr[-1]")
        out = (C.c_double * 2)()
        self._chk(self.lib.pdh_global_error(self.h, sol.ctypes.data, ptr.ctypes.data, p.ctypes.data, ww.ctypes.data,
                                            eu.ctypes.data, eg.ctypes.data, out))
        return float(out[0]), float(out[1])

    def global_error_sums_device(self, d_solution, d_pt_ptr, d_pts, n_points, d_w, d_exact_u, d_exact_grad):
        out = (C.c_double * 2)()
        self._chk(self.lib.pdh_global_error_device(self.h, C.c_void_p(d_solution), C.c_void_p(d_pt_ptr), C.c_void_p(d_pts), n_points,
                                                   C.c_void_p(d_w), C.c_void_p(d_exact_u), C.c_void_p(d_exact_grad), out))
        return float(out[0]), float(out[1])

    def shape_values(self, dim, degree, basis, bbox, pt_ptr, pts):
        """phi_j(x_q) of the box basis (dim, degree, basis) for every box's points: [N][n]."""
        bb = np.ascontiguousarray(bbox, dtype=np.float64).reshape(-1, 2 * dim)
        ptr = np.ascontiguousarray(pt_ptr, dtype=np.int64)
        p = np.ascontiguousarray(pts, dtype=np.float64)
        from .handler import FiniteElement
        n = FiniteElement(dim, degree, basis).n_dofs_per_cell
        out = np.zeros((int(ptr[-1]), n))
        self._chk(self.lib.pdh_shape_values(self.h, dim, degree, basis, bb.shape[0], bb.ctypes.data, ptr.ctypes.data,
                                            p.ctypes.data, out.ctypes.data))
        return out

    def device_values(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.pdh_device_values(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def poison_values(self):
        """Tests: overwrite the resident CSR values with NaN bit patterns (0xFF bytes), so that a value an assembly fails
        to write is seen."""
        ptr, n = self.device_values()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.synchronize()
        if hip.hipMemset(C.c_void_p(ptr), 0xFF, C.c_size_t(8 * n)) != 0 or hip.hipDeviceSynchronize() != 0:
            raise PdhError(-1, "hipMemset of the values failed")

    def set_algorithm(self, alg):
        """'auto' | 'direct' (MFMA contraction over the points) | 'moment' (Legendre moments + sum factorisation)."""
        self._chk(self.lib.pdh_set_algorithm(self.h, {"auto": 0, "direct": 1, "moment": 2, "rows": 4}[alg]))

    def set_exchange_mode(self, mode):
        """'none': owner-computes-rows (no matrix traffic); 'ghost': the reference's scheme - the owner of a cut face ships
        M21 / M22 to the owner of those rows (needs agg_rank in the problem).  Takes effect at the next set_problem."""
        self._chk(self.lib.pdh_set_exchange_mode(self.h, {"none": 0, "ghost": 1}[mode]))

    def exchange_layout(self, n_ranks):
        """(send_count, recv_count): doubles per peer rank of the ghost-block exchange."""
        s = (C.c_int64 * n_ranks)()
        r = (C.c_int64 * n_ranks)()
        self._chk(self.lib.pdh_exchange_layout(self.h, n_ranks, s, r))
        return [int(x) for x in s], [int(x) for x in r]

    def exchange_get_send(self, d_send_ptr):
        self._chk(self.lib.pdh_exchange_get_send(self.h, C.c_void_p(d_send_ptr)))

    def exchange_apply(self, d_recv_ptr):
        self._chk(self.lib.pdh_exchange_apply(self.h, C.c_void_p(d_recv_ptr)))

    def set_stream(self, stream_handle):
        """Launch on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); None: own stream."""
        self._chk(self.lib.pdh_set_stream(self.h, C.c_void_p(stream_handle or 0)))

    def set_overlap(self, on=True):
        """Run the two kernels of a step concurrently on large problems (default) or strictly one after the other."""
        self._chk(self.lib.pdh_set_overlap(self.h, int(on)))

    def algorithm_in_use(self):
        rc = self.lib.pdh_algorithm_in_use(self.h)
        if rc < 0:
            self._chk(rc)
        return {1: "direct", 2: "moment", 3: "mixed", 4: "rows"}[rc]

    def rows_kernel_in_use(self):
        """Which row kernel serves the resident problem (include/polydeal_hip.h: PDH_ROWS_*)."""
        rc = self.lib.pdh_rows_kernel_in_use(self.h)
        if rc < 0:
            self._chk(rc)
        return {0: "none", 1: "pieces", 2: "multi", 3: "streamed", 4: "terms"}[rc]

    def terms_merge_stats(self):
        """Term kernels: {cells, cells_merged, sub_faces, sub_faces_merged} of the owned polytopes (pdh_terms_merge_stats)."""
        out = (C.c_int64 * 4)()
        self._chk(self.lib.pdh_terms_merge_stats(self.h, out))
        return dict(cells=out[0], cells_merged=out[1], sub_faces=out[2], sub_faces_merged=out[3])

    def set_profiling(self, on=True):
        self._chk(self.lib.pdh_set_profiling(self.h, int(on)))

    def kernel_times_ms(self):
        ms = (C.c_float * 2)()
        n = C.c_int()
        self._chk(self.lib.pdh_kernel_times_ms(self.h, ms, C.byref(n)))
        return [float(ms[0]), float(ms[1])], int(n.value)

    def kernel_work(self):
        """MFMA instructions (512 flop each) issued per launch by [k_diag, k_offdiag]."""
        w = (C.c_int64 * 2)()
        self._chk(self.lib.pdh_kernel_work(self.h, w))
        return [int(w[0]), int(w[1])]

    def stats(self):
        st = (C.c_int64 * 8)()
        self._chk(self.lib.pdh_problem_stats(self.h, st))
        keys = ["n_owned_agg", "n_offdiag_items", "n_vq_points", "n_face_side_points", "n_values",
                "dofs_per_cell", "lds_bytes_diag", "lds_bytes_offdiag"]
        return dict(zip(keys, [int(x) for x in st]))

    def close(self):
        if self.h:
            self.lib.pdh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
