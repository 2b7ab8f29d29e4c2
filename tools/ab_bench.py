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
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--fe", default="dgq")
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--block", type=int, default=2)
    ap.add_argument("--alg", default="auto", help="auto | direct | moment | rows (pdh_set_algorithm) for both builds")
    a = ap.parse_args()
    grid, ah, fe = bench.build_handler(pa, a.dim, a.cells, a.block, a.fe, a.degree, a.degree + 1)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    ctxs = {}
    for name, path in (("A", a.lib_a), ("B", a.lib_b)):
        c = pa.Context(0, lib_path=os.path.abspath(path))
        c.set_algorithm(a.alg)
        c.set_problem(flat)
        if hasattr(c.lib, "pdh_set_overlap"):
            c.set_overlap(False)  # per-kernel times of kernels that have the device to themselves
        c.assemble_device()
        c.synchronize()
        ctxs[name] = c
    if ctxs["A"].n_values <= 200_000_000:
        va, vb = ctxs["A"].assemble(), ctxs["B"].assemble()
        print("max |A-B| / max|A| = %.3e" % (np.max(np.abs(va - vb)) / np.max(np.abs(va))))
    times = {"A": [], "B": []}
    for r in range(a.rounds):
        for name in (("A", "B") if r % 2 == 0 else ("B", "A")):
            c = ctxs[name]
            c.set_profiling(True)
            for _ in range(a.steps):
                c.assemble_device()
            (kd, ko), _ = c.kernel_times_ms()
            c.set_profiling(False)
            times[name].append((kd, ko))
    for name in ("A", "B"):
        kd = [t[0] for t in times[name]]
        ko = [t[1] for t in times[name]]
        print("%s: k_diag median %.3f min %.3f | k_offdiag median %.3f min %.3f | total median %.3f ms"
              % (name, statistics.median(kd), min(kd), statistics.median(ko), min(ko),
                 statistics.median([x + y for x, y in times[name]])))


if __name__ == "__main__":
    main()
