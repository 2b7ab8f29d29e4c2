#!/usr/bin/env python3
"""Term kernel (pdh_terms.h) against the kinds of pdh_rows.h / the two-kernel forms on the bench mesh, same process, same
device: ms per assembly (HIP events), which kernel ran, largest difference between the two results.
usage: terms_time.py [cells=64] [cases: dgp3,dgq2,dgp2,dgq1,dgp1] [grown]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cases = (sys.argv[2] if len(sys.argv) > 2 else "dgp3,dgq2,dgp2,dgq1,dgp1").split(",")
grown = len(sys.argv) > 3 and sys.argv[3] == "grown"
for cs in cases:
    basis, p = cs[:3], int(cs[3])
    grid, ah, fe = bench.build_handler(pa, 3, cells, 2, basis, p, p + 1, grown=grown)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    res = {}
    for env in ("1", "0"):
        os.environ["PDH_TERMS"] = env
        ctx = pa.Context(0)
        ctx.set_overlap(False)
        ctx.set_problem(flat)
        used = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
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
        cs_ = ctx.checksum()
        res[env] = (statistics.median(ts), min(ts), used, vals, cs_)
        ctx.close()
    a, b = res["1"], res["0"]
    diff = (np.max(np.abs(a[3] - b[3])) / np.max(np.abs(b[3]))) if a[3] is not None and b[3] is not None else float("nan")
    print("%s%d %s: terms %.3f ms (min %.3f) %s | PDH_TERMS=0 %.3f ms (min %.3f) %s | rel diff %.2e | sums %.12e %.12e"
          % (basis, p, "grown" if grown else "block", a[0], a[1], a[2], b[0], b[1], b[2], diff, a[4]["sum"], b[4]["sum"]), flush=True)
