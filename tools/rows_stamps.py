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
# Balance between the persistent waves: wave g works through slots g, g + G, g + 2 G, ... (G = grid size); the stamps of one
# wave come from one counter, so its own span (first start -> last end) is meaningful, absolute times of different waves are not
G = int(os.environ.get("PDH_ROWS_GRID", "0")) or min(n, 8 * 256)
if basis == "dgq" and degree == 3:
    # (this kind hands its polytopes out through a device-wide counter since round 3: a slot no longer tells which wave worked on
    # it; the balance figures below belong to the kinds with the static stride - profiles/r03_rows_stamps_static_balance.txt holds
    # the last measurement of FE_DGQ(3) with the stride: spans 0.81 .. 1.15 of the mean)
    sys.exit(0)
per_wave = []
for g in range(min(G, n)):
    sl = np.arange(g, n, G)
    per_wave.append((out[sl[-1], 6] - out[sl[0], 0], len(sl), float(tot[sl].sum())))
pw = np.array(per_wave, dtype=np.float64)
print("per wave (%d waves, %d-%d polytopes each): span min %.0f  mean %.0f  median %.0f  p95 %.0f  max %.0f ticks; busy (sum of polytope lifetimes) / span: mean %.3f"
      % (len(pw), pw[:, 1].min(), pw[:, 1].max(), pw[:, 0].min(), pw[:, 0].mean(), np.median(pw[:, 0]), np.percentile(pw[:, 0], 95), pw[:, 0].max(),
         float((pw[:, 2] / pw[:, 0]).mean())))
by_xcd = [pw[x::8, 0].mean() for x in range(8)]
print("mean span by blockIdx %% 8 (XCD under round-robin placement): " + " ".join("%.0f" % v for v in by_xcd))
