#!/usr/bin/env python3
"""Development check + timing of the workgroup-per-polytope kernel for FE_DGQ(3) (pdh_terms_wg.h; PDH_TERMS_DGQ3=1) - parity with the
oracle on small block / staircase problems in both CSR layouts, then ms per assembly on the bench mesh against pdh_rows.h.
usage: terms_wg_check.py [check|time|both] [cells=64] [grown]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import polydeal_amd as pa  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 64
grown = len(sys.argv) > 3 and sys.argv[3] == "grown"
os.environ["PDH_TERMS_DGQ3"] = "1"
bad = 0
if mode in ("check", "both"):
    from flatten_oracle import flatten
    from oracle import polydeal_oracle as po
    from parity import assert_parity_ah
    from test_gpu_parity import random_agglomeration, variant

    for kind, n, per, vname, diag_first in (("block", 4, 2, "poisson", True), ("block", 4, 2, "dr", False), ("block", 2, 2, "adm", True),
                                            ("block", 4, 1, "test", True), ("block", 6, 2, "poisson", True), ("grown", 4, 4, "poisson", True),
                                            ("grown", 6, 6, "dr", False), ("grown", 6, 3, "minsip", True), ("block", 6, 3, "adm", False),
                                            ("grown", 8, 8, "poisson", True)):
        fe = po.FE_DGQ(3, 3)
        grid = po.subdivided_hyper_cube(3, n, 0.0, 1.0)
        ah = po.AgglomerationHandler(grid)
        groups = po.block_agglomerates(grid, per) if kind == "block" else random_agglomeration(grid, max(2, grid.n_cells // per), np.random.default_rng(n + per))
        for g in groups:
            ah.define_agglomerate(g)
        ah.initialize_fe_values(4, 4)
        ah.distribute_agglomerated_dofs(fe)
        var = variant(vname, fe)
        kw = flatten(ah, var, diag_first=diag_first)
        ref = po.assemble_csr(ah, var, diag_first=diag_first)[2]
        ctx = pa.Context(0)
        ctx.set_problem(pa.Problem(**kw))
        used = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
        v = ctx.assemble()
        ctx.close()
        err = np.max(np.abs(v - ref)) / np.max(np.abs(ref))
        ok = True
        try:
            assert_parity_ah(v, ref, ah, diag_first, what="terms wg")
        except AssertionError as e:
            ok, bad = False, bad + 1
            print("   ", str(e)[:300])
        print("dgq3 %-5s n=%d per=%d %-7s diag_first=%d W=%s: %s err %.2e %s" % (kind, n, per, vname, diag_first, os.environ.get("PDH_TERMS_WG_WAVES", "4"),
                                                                              used, err, "ok" if ok else "FAIL"), flush=True)
    print("failures:", bad)
if mode in ("time", "both") and not bad:
    import bench

    grid, ah, fe = bench.build_handler(pa, 3, cells, 2, "dgq", 3, 4, grown=grown)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    res = {}
    for env in ("1", "0"):
        os.environ["PDH_TERMS_DGQ3"] = env
        ctx = pa.Context(0)
        ctx.set_overlap(False)
        ctx.set_problem(flat)
        used = ctx.algorithm_in_use(), ctx.rows_kernel_in_use()
        for _ in range(3):
            ctx.assemble_device()
        ctx.synchronize()
        ts = []
        for _ in range(6):
            ctx.set_profiling(True)
            for _ in range(4):
                ctx.assemble_device()
            (k0, k1), _ = ctx.kernel_times_ms()
            ctx.set_profiling(False)
            ts.append(k0 + k1)
        res[env] = (statistics.median(ts), min(ts), used, ctx.checksum())
        ctx.close()
    a, b = res["1"], res["0"]
    print("dgq3 %s W=%s: terms_wg %.3f ms (min %.3f) %s | pdh_rows %.3f ms (min %.3f) %s | sums %.10e %.10e"
          % ("grown" if grown else "block", os.environ.get("PDH_TERMS_WG_WAVES", "4"), a[0], a[1], a[2], b[0], b[1], b[2], a[3]["sum"], b[3]["sum"]), flush=True)
sys.exit(1 if bad else 0)
