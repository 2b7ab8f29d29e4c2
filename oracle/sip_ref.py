"""Loader for the plain-C restatement oracle/sip_ref.c (TEST INFRASTRUCTURE: checker + timed CPU baseline).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "sip_ref.c")
_OUT = os.path.join(_HERE, "_build", "libsip_ref.so")
_OUT_FAST = os.path.join(_HERE, "_build", "libsip_ref_fast.so")

_FIELDS = [
    ("dim", C.c_int32), ("degree", C.c_int32), ("basis", C.c_int32), ("n_agg", C.c_int32),
    ("n_faces", C.c_int32), ("n_rows", C.c_int32), ("diag_first", C.c_int32), ("reserved", C.c_int32),
    ("reaction_c", C.c_double),
    ("bbox", C.c_void_p), ("dof_offset", C.c_void_p),
    ("vq_ptr", C.c_void_p), ("vq_x", C.c_void_p), ("vq_w", C.c_void_p),
    ("face_in", C.c_void_p), ("face_out", C.c_void_p), ("fq_ptr", C.c_void_p),
    ("fq_x", C.c_void_p), ("fq_n", C.c_void_p), ("fq_w", C.c_void_p), ("fq_w_out", C.c_void_p),
    ("face_sigma", C.c_void_p), ("rowptr", C.c_void_p), ("colind", C.c_void_p),
]
_DT = {
    "bbox": np.float64, "dof_offset": np.int32, "vq_ptr": np.int64, "vq_x": np.float64, "vq_w": np.float64,
    "face_in": np.int32, "face_out": np.int32, "fq_ptr": np.int64, "fq_x": np.float64, "fq_n": np.float64,
    "fq_w": np.float64, "fq_w_out": np.float64, "face_sigma": np.float64, "rowptr": np.int64, "colind": np.int32,
}


class _Problem(C.Structure):
    _fields_ = _FIELDS


def build(force=False):
    """gcc -O2 -fopenmp (deal.II Release default optimisation level)."""
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _OUT, "-lm"])
    if force or not os.path.exists(_OUT_FAST) or os.path.getmtime(_OUT_FAST) < os.path.getmtime(_SRC):
        # the "honest best-effort" variant: same source, full optimisation.  x86-64-v3 (AVX2 + FMA) rather than
        # -march=native: the library is built in one container and timed on another host (a native build for a newer CPU
        # would die there with SIGILL, which no try/except catches)
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _OUT_FAST, "-lm"])
    return _OUT


_lib = None
_lib_fast = None


def lib_fast():
    global _lib_fast
    if _lib_fast is None:
        build()
        _lib_fast = C.CDLL(_OUT_FAST)
        _lib_fast.sipref_assemble_fast.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_int, C.c_int, C.c_int]
    return _lib_fast


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sipref_assemble.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_int, C.c_int, C.c_int]
    return _lib


def max_threads():
    return lib().sipref_max_threads()


def assemble(kw, a_begin=0, a_end=None, nthreads=1, fast=False):
    """kw: the keyword dict of a pdh_problem (as tests/flatten_oracle.flatten or FlatView.arrays() + scalars).
    Returns (values, seconds)."""
    p = _Problem()
    keep = []
    for name, ct in _FIELDS:
        if name in _DT:
            a = kw.get(name)
            if a is None:
                setattr(p, name, None)
            else:
                a = np.ascontiguousarray(a, dtype=_DT[name])
                keep.append(a)
                setattr(p, name, a.ctypes.data)
        elif name != "reserved":
            setattr(p, name, kw[name] if name != "reaction_c" else float(kw.get("reaction_c", 0.0)))
    nnz = int(np.asarray(kw["rowptr"])[-1])
    values = np.zeros(nnz)
    a_end = p.n_agg if a_end is None else a_end
    t0 = time.perf_counter()
    if fast:
        rc = lib_fast().sipref_assemble_fast(C.byref(p), values.ctypes.data, a_begin, a_end, nthreads)
    else:
        rc = lib().sipref_assemble(C.byref(p), values.ctypes.data, a_begin, a_end, nthreads)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("sipref_assemble failed with %d" % rc)
    return values, dt
