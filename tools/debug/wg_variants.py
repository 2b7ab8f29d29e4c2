#!/usr/bin/env python3
"""Times several builds of the library on the FE_DGQ(3) bench mesh with the workgroup kernel (PDH_TERMS_DGQ3=1), interleaved,
two contexts per build: usage wg_variants.py name=path ..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

os.environ["PDH_TERMS_DGQ3"] = os.environ.get("PDH_TERMS_DGQ3", "1")
libs = [a.split("=", 1) for a in sys.argv[1:]]
grid, ah, fe = bench.build_handler(pa, 3, 64, 2, "dgq", 3, 4)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
ctxs = {}
for rep in range(2):
    for name, path in libs:
        c = pa.Context(0, lib_path=os.path.abspath(path))
        c.set_overlap(False)
        c.set_problem(flat)
        c.assemble_device()
        c.synchronize()
        ctxs.setdefault(name, []).append(c)
times = {n: [] for n, _ in libs}
for r in range(6):
    for name, _ in (libs if r % 2 == 0 else libs[::-1]):
        for c in ctxs[name]:
            c.set_profiling(True)
            for _ in range(3):
                c.assemble_device()
            (k0, k1), _ = c.kernel_times_ms()
            c.set_profiling(False)
            times[name].append(k0 + k1)
for name, _ in libs:
    c = ctxs[name][0]
    print("%-12s %s median %.3f ms  min %.3f  max %.3f" % (name, c.rows_kernel_in_use(), statistics.median(times[name]), min(times[name]), max(times[name])), flush=True)
