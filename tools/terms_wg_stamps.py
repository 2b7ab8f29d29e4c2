#!/usr/bin/env python3
"""Phase timing of the workgroup-per-polytope kernel (pdh_terms_wg.h, W = 4) from in-kernel cycle-counter stamps
(library built with -DPDHT_STAMP):  python tools/terms_wg_stamps.py build/tstamp/libpolydeal_hip.so [cells] [grown]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

os.environ["PDH_TERMS_DGQ3"] = "1"
lib_path = os.path.abspath(sys.argv[1])
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 64
grown = len(sys.argv) > 3 and sys.argv[3] == "grown"
grid, ah, fe = bench.build_handler(pa, 3, cells, 2, "dgq", 3, 4, grown=grown)
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
st = out.reshape(n, 4, 4).astype(np.float64)
life = st[:, :, 3] - st[:, :, 0]
print("FE_DGQ(3) %s, W = 4: %d polytopes; mean lifetime of a wave %.0f cycles" % ("grown" if grown else "blocks", n, life.mean()))
names = ["A: loads + lane tasks + barriers", "B: own block (terms of cells and sub-faces)", "B: pieces (coupling terms + all row stores)"]
for k, nm in enumerate(names):
    d = st[:, :, k + 1] - st[:, :, k]
    print("%-48s mean %8.0f (%4.1f %%)   by wave: %s" % (nm, d.mean(), 100 * d.mean() / life.mean(), " ".join("%.0f" % d[:, w].mean() for w in range(4))))
