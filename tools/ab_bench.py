#!/usr/bin/env python3
"""A/B two builds of libpolydeal_hip.so in ONE process on ONE device, interleaved rounds (the only way to see
differences of a few percent: boxes and runs differ by that much).  Usage (on the GPU box):
    python tools/ab_bench.py --lib-a polydeal_amd/lib/libpolydeal_hip.so --lib-b build/variant/libpolydeal_hip.so
Prints median / min kernel times of both builds and checks that their matrices agree."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib-a", required=True)
    ap.add_argument("--lib-b", required=True)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--contexts", type=int, default=1, help="contexts per build (created alternately)")
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--fe", default="dgq")
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--block", type=int, default=2)
    ap.add_argument("--grown", action="store_true", help="irregular agglomerates grown over the cell graph (MULTI row kernel)")
    ap.add_argument("--alg", default="auto", help="auto | direct | moment | rows (pdh_set_algorithm) for both builds")
    a = ap.parse_args()
    grid, ah, fe = bench.build_handler(pa, a.dim, a.cells, a.block, a.fe, a.degree, a.degree + 1, grown=a.grown)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    # (several contexts per build, created alternately: the same code runs up to 3 % apart in two contexts of one process -
    # where the allocator puts the 7 GB of values matters - so one context per build cannot resolve differences of that size)
    ctxs = {"A": [], "B": []}
    for k in range(a.contexts):
        for name, path in (("A", a.lib_a), ("B", a.lib_b)):
            c = pa.Context(0, lib_path=os.path.abspath(path))
            c.set_algorithm(a.alg)
            c.set_problem(flat)
            if hasattr(c.lib, "pdh_set_overlap"):
                c.set_overlap(False)  # per-kernel times of kernels that have the device to themselves
            c.assemble_device()
            c.synchronize()
            ctxs[name].append(c)
    if ctxs["A"][0].n_values <= 200_000_000:
        va, vb = ctxs["A"][0].assemble(), ctxs["B"][0].assemble()
        print("max |A-B| / max|A| = %.3e" % (np.max(np.abs(va - vb)) / np.max(np.abs(va))))
    times = {"A": [[] for _ in range(a.contexts)], "B": [[] for _ in range(a.contexts)]}
    for r in range(a.rounds):
        for k in range(a.contexts):
            for name in (("A", "B") if (r + k) % 2 == 0 else ("B", "A")):
                c = ctxs[name][k]
                c.set_profiling(True)
                for _ in range(a.steps):
                    c.assemble_device()
                (kd, ko), _ = c.kernel_times_ms()
                c.set_profiling(False)
                times[name][k].append((kd, ko))
    for name in ("A", "B"):
        allt = [t for k in range(a.contexts) for t in times[name][k]]
        kd = [t[0] for t in allt]
        ko = [t[1] for t in allt]
        per_ctx = " ".join("%.3f" % statistics.median([t[0] for t in times[name][k]]) for k in range(a.contexts))
        print("%s: k_diag median %.3f min %.3f | k_offdiag median %.3f min %.3f | total median %.3f ms%s"
              % (name, statistics.median(kd), min(kd), statistics.median(ko), min(ko),
                 statistics.median([x + y for x, y in allt]),
                 (" | k_diag median per context: " + per_ctx) if a.contexts > 1 else ""))


if __name__ == "__main__":
    main()
