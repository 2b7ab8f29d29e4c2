#!/usr/bin/env python3
"""Term kernels with and without merged cells / sub-faces (PDH_TERMS_MERGE), same process, same device: ms per assembly (HIP events), the
numbers of cells and sub-faces summed over, largest difference between the two results; for FE_DGQ(3) also pdh_rows.h.
usage: merge_time.py [cells=64] [cases: dgq3,dgp3,dgq2,dgp2,dgq1,dgp1] [block=2] [cartesian description: 0|1] [grown agglomerates: 0|1]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cases = (sys.argv[2] if len(sys.argv) > 2 else "dgq3,dgp3,dgq2,dgp2,dgq1,dgp1").split(",")
block = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cart = len(sys.argv) > 4 and sys.argv[4] == "1"
grown = len(sys.argv) > 5 and sys.argv[5] == "1"
for cs in cases:
    basis, p = cs[:3], int(cs[3])
    grid, ah, fe = bench.build_handler(pa, 3, cells, block, basis, p, p + 1, grown=grown)
    var = pa.SipVariant.poisson_example(fe)
    flat = ah.flatten_cartesian(var, True, False) if cart else ah.flatten(var, True, False)
    res = {}
    forms = [("merged", "1", "1"), ("as given", "0", "1")] + ([("pdh_rows.h", "1", "0")] if (basis == "dgq" and p == 3 and not cart) else [])
    for name, mg, q3 in forms:
        os.environ["PDH_TERMS_MERGE"] = mg
        os.environ["PDH_TERMS_DGQ3"] = q3
        ctx = pa.Context(0)
        ctx.set_overlap(False)
        try:
            ctx.set_problem(flat)
        except pa.PdhError as e:
            print("%s%d %s: %s" % (basis, p, name, str(e)[:100]), flush=True)
            ctx.close()
            continue
        used = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
        st = ctx.terms_merge_stats()
        for _ in range(3):
            ctx.assemble_device()
        ctx.synchronize()
        ts = []
        for _ in range(5):
            ctx.set_profiling(True)
            for _ in range(4):
                ctx.assemble_device()
            (k0, k1), _ = ctx.kernel_times_ms()
            ctx.set_profiling(False)
            ts.append(k0 + k1)
        vals = ctx.assemble() if ctx.n_values <= 300_000_000 else None
        res[name] = (statistics.median(ts), min(ts), used, vals, ctx.checksum(), st)
        ctx.close()
    ref = res.get("as given")
    for name, r in res.items():
        diff = (np.max(np.abs(r[3] - ref[3])) / np.max(np.abs(ref[3]))) if (ref and r[3] is not None and ref[3] is not None) else float("nan")
        print("%s%d b=%d %-10s %.3f ms (min %.3f) %s cells %d -> %d sub-faces %d -> %d | rel diff to as-given %.2e | sum %.12e"
              % (basis, p, block, name, r[0], r[1], r[2], r[5]["cells"], r[5]["cells_merged"], r[5]["sub_faces"], r[5]["sub_faces_merged"],
                 diff, r[4]["sum"]), flush=True)
