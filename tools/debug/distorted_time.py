#!/usr/bin/env python3
"""The general (non-Cartesian) path on the distorted bench mesh: per-kernel ms (HIP events, kernels serialised) and ms per assembly with the
two kernels overlapped, for both CSR layouts and every form that applies.
usage: distorted_time.py [cells=64] [cases: dgq3,dgp3,dgq2] [distort=0.1]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402
import torch  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cases = (sys.argv[2] if len(sys.argv) > 2 else "dgq3,dgp3,dgq2").split(",")
distort = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
for cs in cases:
    basis, p = cs[:3], int(cs[3])
    grid, ah, fe = bench.build_handler(pa, 3, cells, 2, basis, p, p + 1, distort=distort)
    for diag_first in (True, False):
        flat = ah.flatten(pa.SipVariant.poisson_example(fe), diag_first, False)
        for alg in ("auto", "moment", "direct"):
            ctx = pa.Context(0)
            try:
                ctx.set_algorithm(alg)
                ctx.set_problem(flat)
            except pa.PdhError as e:
                print("%s%d diag_first=%d %s: %s" % (basis, p, diag_first, alg, str(e)[:80]), flush=True)
                ctx.close()
                continue
            used = ctx.algorithm_in_use()
            if alg != "auto" and used != alg:
                ctx.close()
                continue
            out = {}
            for ov in (False, True):
                ctx.set_overlap(ov)
                for _ in range(3):
                    ctx.assemble_device()
                ctx.synchronize()
                ts, ks = [], []
                for _ in range(5):
                    ctx.set_profiling(True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(4):
                        ctx.assemble_device()
                    ctx.synchronize()
                    ts.append((time.perf_counter() - t0) / 4 * 1e3)
                    (k0, k1), _ = ctx.kernel_times_ms()
                    ctx.set_profiling(False)
                    ks.append((k0, k1))
                out[ov] = (statistics.median(ts), statistics.median(k[0] for k in ks), statistics.median(k[1] for k in ks))
            c = ctx.checksum()
            print("%s%d diag_first=%d asked %-6s ran %-6s: serial %.3f ms (k0 %.3f + k1 %.3f) | overlapped %.3f ms | sum %.10e nonfinite %d"
                  % (basis, p, diag_first, alg, used, out[False][0], out[False][1], out[False][2], out[True][0], c["sum"], c["non_finite"]),
                  flush=True)
            ctx.close()
