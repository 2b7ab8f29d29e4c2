#!/usr/bin/env python3
"""Development check of pdh_set_problem_cartesian: values against the points-based path and (small cases) the oracle; set-up and
first-assembly times of both paths on the bench mesh.  usage: cart_check.py [check|time|both]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
bad = 0
if mode in ("check", "both"):
    for basis, p in (("dgp", 3), ("dgq", 3), ("dgq", 2), ("dgp", 1)):
        for cells, grown, vname, diag_first in ((4, False, "poisson", True), (6, True, "dr", False), (8, True, "adm", True), (6, False, "minsip", True)):
            grid, ah, fe = bench.build_handler(pa, 3, cells, 2, basis, p, p + 1, grown=grown)
            var = {"poisson": pa.SipVariant.poisson_example(fe), "dr": pa.SipVariant.diffusion_reaction(fe),
                   "adm": pa.SipVariant.assemble_dg_matrix(), "minsip": pa.SipVariant.minimal_sip_example()}[vname]
            ref_flat = ah.flatten(var, diag_first, False)
            c0 = pa.Context(0)
            c0.set_problem(ref_flat)
            v0 = c0.assemble()
            k0 = c0.rows_kernel_in_use()
            c0.close()
            cf = ah.flatten_cartesian(var, diag_first, False)
            c1 = pa.Context(0)
            c1.set_problem(cf)
            v1 = c1.assemble()
            k1 = c1.rows_kernel_in_use()
            c1.close()
            err = np.max(np.abs(v0 - v1)) / np.max(np.abs(v0))
            ok = err <= 1e-13
            bad += 0 if ok else 1
            print("%s%d cells=%d grown=%d %-7s diag_first=%d: points %s | cartesian %s | rel diff %.2e %s" % (basis, p, cells, grown, vname, diag_first, k0, k1, err, "ok" if ok else "FAIL"), flush=True)
    print("failures:", bad)
if mode in ("time", "both") and not bad:
    ctx0 = pa.Context(0)
    for basis in ("dgq", "dgp"):
        for rep in range(2):
            grid, ah, fe = bench.build_handler(pa, 3, 64, 2, basis, 3, 4)
            var = pa.SipVariant.poisson_example(fe)
            t0 = time.time()
            flat = ah.flatten(var, True, False)
            t1 = time.time()
            c = pa.Context(0)
            c.set_problem(flat)
            c.assemble_device()
            c.synchronize()
            t2 = time.time()
            s0 = c.checksum()["sum"]
            c.close()
            del flat
            t3 = time.time()
            cf = ah.flatten_cartesian(var, True, False)
            t4 = time.time()
            c = pa.Context(0)
            c.set_problem(cf)
            c.assemble_device()
            c.synchronize()
            t5 = time.time()
            s1 = c.checksum()["sum"]
            k = c.rows_kernel_in_use()
            c.close()
            print("%s rep %d: points: flatten %.3f + set_problem/assemble %.3f = %.3f s | cartesian: flatten %.3f + set_problem/assemble %.3f = %.3f s (%s) | sums %.10e %.10e"
                  % (basis, rep, t1 - t0, t2 - t1, t2 - t0, t4 - t3, t5 - t4, t5 - t3, k, s0, s1), flush=True)
sys.exit(1 if bad else 0)
