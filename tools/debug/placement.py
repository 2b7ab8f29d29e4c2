#!/usr/bin/env python3
"""How much does the kernel time depend on WHERE the problem's arrays sit in device memory?  Creates N contexts of the
same library on the same problem one after the other, times each, frees them, and does it again.
    python tools/debug/placement.py [N=6] [rounds=2]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
grid, ah, fe = bench.build_handler(pa, 3, 64, 2, "dgq", 3, 4)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)


def timed(c, reps=5):
    ts = []
    for _ in range(reps):
        c.set_profiling(True)
        for _ in range(3):
            c.assemble_device()
        (kd, _), _ = c.kernel_times_ms()
        c.set_profiling(False)
        ts.append(kd)
    return statistics.median(ts)


for r in range(rounds):
    ctxs = []
    for k in range(n_ctx):
        c = pa.Context(0)
        c.set_problem(flat)
        c.assemble_device()
        c.synchronize()
        ctxs.append(c)
    line = []
    for k, c in enumerate(ctxs):
        ptr, n = c.device_values()
        line.append("%.3f ms @ 0x%x" % (timed(c), ptr))
    print("round %d: %s" % (r, " | ".join(line)), flush=True)
    # a second pass over the same contexts: is the time a property of the context?
    print("   again: %s" % " | ".join("%.3f" % timed(c) for c in ctxs), flush=True)
    for c in ctxs:
        c.close()
    del ctxs
