#!/usr/bin/env python3
"""Several builds of the library on the bench mesh for the small elements (term kernel), interleaved, two contexts per build; the
first build's values are the yardstick for the others.  usage: terms_variants.py cases name=path ...   (cases: dgp3,dgq2,...[,grown])"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

cases = sys.argv[1].split(",")
grown = "grown" in cases
cases = [c for c in cases if c != "grown"]
libs = [a.split("=", 1) for a in sys.argv[2:]]
for cs in cases:
    basis, p = cs[:3], int(cs[3])
    grid, ah, fe = bench.build_handler(pa, 3, 64, 2, basis, p, p + 1, grown=grown)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    ctxs, vals = {}, {}
    for rep in range(2):
        for name, path in libs:
            c = pa.Context(0, lib_path=os.path.abspath(path))
            c.set_overlap(False)
            c.set_problem(flat)
            c.assemble_device()
            c.synchronize()
            ctxs.setdefault(name, []).append(c)
    for name, _ in libs:
        c = ctxs[name][0]
        vals[name] = c.assemble() if c.n_values <= 300_000_000 else None
    times = {n: [] for n, _ in libs}
    for r in range(6):
        for name, _ in (libs if r % 2 == 0 else libs[::-1]):
            for c in ctxs[name]:
                c.set_profiling(True)
                for _ in range(4):
                    c.assemble_device()
                (k0, k1), _ = c.kernel_times_ms()
                c.set_profiling(False)
                times[name].append(k0 + k1)
    ref = vals[libs[0][0]]
    for name, _ in libs:
        d = float(np.max(np.abs(vals[name] - ref)) / np.max(np.abs(ref))) if ref is not None and vals[name] is not None else float("nan")
        print("%s%d %-6s %-10s %s median %.3f ms  min %.3f  max %.3f  | rel diff to %s %.1e" % (basis, p, "grown" if grown else "block", name, ctxs[name][0].rows_kernel_in_use(),
              statistics.median(times[name]), min(times[name]), max(times[name]), libs[0][0], d), flush=True)
    for cl in ctxs.values():
        for c in cl:
            c.close()
