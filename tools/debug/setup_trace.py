#!/usr/bin/env python3
"""Where the set-up of the bench problem goes: handler, flatten, pdh_set_problem (PDH_TRACE_SETUP=1 prints its phases)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PDH_TRACE_SETUP"] = "1"
import bench  # noqa: E402
import polydeal_amd as pa  # noqa: E402

basis = sys.argv[1] if len(sys.argv) > 1 else "dgq"
ctx0 = pa.Context(0)  # HIP runtime + device context: not problem set-up
for rep in range(2):
    t0 = time.time()
    grid, ah, fe = bench.build_handler(pa, 3, 64, 2, basis, 3, 4)
    t1 = time.time()
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    t2 = time.time()
    ctx = pa.Context(0)
    ctx.set_problem(flat)
    ctx.synchronize()
    t3 = time.time()
    ctx.assemble_device()
    ctx.synchronize()
    t4 = time.time()
    print("rep %d %s: handler %.3f flatten %.3f set_problem %.3f first assembly %.4f s (%s)" % (rep, basis, t1 - t0, t2 - t1, t3 - t2, t4 - t3, ctx.rows_kernel_in_use()), flush=True)
    ctx.close()
