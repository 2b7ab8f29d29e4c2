#!/usr/bin/env python3
"""Phase timing of the term kernel (pdh_terms.h) from in-kernel cycle-counter stamps (a library built with -DPDHT_STAMP):
    make -C polydeal_amd/csrc DEFS=-DPDHT_STAMP OUT=../../build/tstamp/libpolydeal_hip.so BUILD=../../build/tstamp/obj
    python tools/terms_stamps.py build/tstamp/libpolydeal_hip.so [cells] [dgq|dgp] [degree] [grown]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

lib_path = os.path.abspath(sys.argv[1])
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 64
basis = sys.argv[3] if len(sys.argv) > 3 else "dgp"
degree = int(sys.argv[4]) if len(sys.argv) > 4 else 3
grown = len(sys.argv) > 5 and sys.argv[5] == "grown"
grid, ah, fe = bench.build_handler(pa, 3, cells, 2, basis, degree, degree + 1, grown=grown)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
ctx = pa.Context(0, lib_path=lib_path)
ctx.set_problem(flat)
assert ctx.rows_kernel_in_use() == "terms"
for _ in range(3):
    ctx.assemble_device()
ctx.synchronize()
n = ctx.stats()["n_owned_agg"]
out = np.zeros((n, 16), dtype=np.int64)
ctx.lib.pdh_debug_rows_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert ctx.lib.pdh_debug_rows_stamps(ctx.h, out.ctypes.data) == 0
d = np.diff(out[:, :6], axis=1).astype(np.float64)
tot = (out[:, 5] - out[:, 0]).astype(np.float64)
names = ["record + digit table", "A: (sub-face, tangential direction) tasks", "A: normal-direction and cell tasks", "B1: diagonal block",
         "B2: rows (terms of the coupling columns + stores)"]
print("%s(%d) %s: polytopes %d, mean lifetime of a wave %.0f cycles (median %.0f)" % (basis, degree, "grown" if grown else "blocks", n, tot.mean(), np.median(tot)))
for k, nm in enumerate(names):
    print("%-52s mean %9.0f  median %9.0f  (%4.1f %%)" % (nm, d[:, k].mean(), np.median(d[:, k]), 100 * d[:, k].mean() / tot.mean()))
