#!/usr/bin/env python3
"""Phase timing of the row kernel from in-kernel s_memtime stamps (a library built with -DPDHR_STAMP):
    make -C polydeal_amd/csrc DEFS=-DPDHR_STAMP OUT=../../build/stamp/libpolydeal_hip.so BUILD=../../build/stamp/obj
    python tools/rows_stamps.py build/stamp/libpolydeal_hip.so
Prints the mean wave cycles between phase boundaries (100 MHz s_memtime ticks are NOT used: readcyclecounter = shader clock)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

lib_path = os.path.abspath(sys.argv[1])
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 64
basis = sys.argv[3] if len(sys.argv) > 3 else "dgq"
degree = int(sys.argv[4]) if len(sys.argv) > 4 else 3
grown = len(sys.argv) > 5 and sys.argv[5] == "grown"  # irregular agglomerates (MULTI instantiation of the kernel)
if basis != "dgq" or degree != 3:
    os.environ["PDH_TERMS"] = "0"  # (the streamed kinds of pdh_rows.h: the term kernel has its own stamps, tools/terms_stamps.py)
grid, ah, fe = bench.build_handler(pa, 3, cells, 2, basis, degree, degree + 1, grown=grown)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
ctx = pa.Context(0, lib_path=lib_path)
ctx.set_problem(flat)
assert ctx.algorithm_in_use() == "rows"
for _ in range(3):
    ctx.assemble_device()
ctx.synchronize()
n = ctx.stats()["n_owned_agg"]
out = np.zeros((n, 16), dtype=np.int64)
ctx.lib.pdh_debug_rows_stamps.argtypes = [C.c_void_p, C.c_void_p]
rc = ctx.lib.pdh_debug_rows_stamps(ctx.h, out.ctypes.data)
assert rc == 0
d = np.diff(out[:, :7], axis=1).astype(np.float64)
names = ["prologue (tables, face table)", "P1 volume", "P2 faces", "P3 carry", "P4 diagonal block", "P5 coupling blocks"]
tot = (out[:, 6] - out[:, 0]).astype(np.float64)
print("waves %d, mean lifetime %.0f ticks (median %.0f)" % (n, tot.mean(), np.median(tot)))
for k, nm in enumerate(names):
    print("%-32s mean %9.0f  median %9.0f  (%4.1f %%)" % (nm, d[:, k].mean(), np.median(d[:, k]), 100 * d[:, k].mean() / tot.mean()))
tensor = out[:, 12].any() or out[:, 13].any()
if tensor:  # P2 ran on verified tensor sub-face rules (slots 12, 13); slots 8 .. 11 belong to the general-point path
    for k, nm in ((12, "P2 (tensor rules): lane tasks"), (13, "P2 (tensor rules): per-face sums + expansion")):
        print("%-40s mean %9.0f" % (nm, out[:, k].mean()))
    if grown:  # MULTI instantiation: slots 8, 9 time the two parts of P5
        for k, nm in ((8, "P5 (MULTI): S and C of the plane entries"), (9, "P5 (MULTI): carries + row stores")):
            print("%-40s mean %9.0f" % (nm, out[:, k].mean()))
else:
    for k, nm in ((8, "P2: issue of the next chunk's loads"), (9, "P2: record phase"), (10, "P2: MFMA steps"), (11, "P2: flush + expansion")):
        print("%-40s mean %9.0f" % (nm, out[:, k].mean()))
if basis == "dgq" and degree == 3:
    for k, nm in ((7, "P4: stage 1 (a2, VALU)"), (14, "P4: stages 2 + 3 (MFMA, three groups)"), (15, "P4: rows of the own block")):
        print("%-40s mean %9.0f" % (nm, out[:, k].mean()))
else:
    for k, nm in ((7, "P4 (streamed kinds): stage 1"), (14, "P4 (streamed kinds): stage 2"), (15, "P4 (streamed kinds): stage 3")):
        print("%-40s mean %9.0f" % (nm, out[:, k].mean()))
# (Round 3 printed a "balance between the persistent waves" block here that assumed a grid of min(n, 8 x 256) waves with a static
# stride.  FE_DGQ(3) hands its polytopes out through a device-wide counter (a slot no longer tells which wave worked on it) and the
# streamed kinds run with the grid the occupancy query gives (7 x 256 for FE_AggloDGP(3)), so the slots it grouped belonged to
# different waves: per-wave spans of -3e11 .. +3e11 ticks.  Removed; profiles/r03_rows_stamps_static_balance.txt holds the last valid
# measurement of the static stride - spans 0.81 .. 1.15 of the mean.  The phase shares above do not depend on it.)
