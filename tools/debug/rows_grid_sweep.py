#!/usr/bin/env python3
"""FE_DGQ(3) row kernel (pdh_rows.h) on the bench mesh with more workgroups than are resident at once (PDH_ROWS_WAVES_PER_CU: the
launcher's diagnostic override; 128 = one workgroup per polytope, the device-wide counter then hands out nothing): does the
hardware dispatcher balance better than the persistent waves' own counter?  One process per setting (the override is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import os, sys, statistics
sys.path.insert(0, %r)
import bench, polydeal_amd as pa
grid, ah, fe = bench.build_handler(pa, 3, 64, 2, 'dgq', 3, 4)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
ts = []
for rep in range(3):
    c = pa.Context(0); c.set_overlap(False); c.set_problem(flat)
    for _ in range(3): c.assemble_device()
    c.synchronize()
    for r in range(4):
        c.set_profiling(True)
        for _ in range(4): c.assemble_device()
        (k0, k1), _ = c.kernel_times_ms(); c.set_profiling(False); ts.append(k0 + k1)
    cs = c.checksum(); c.close()
print('waves/CU=%%s: median %%.3f ms min %%.3f max %%.3f (sum %%.10e)' %% (os.environ.get('PDH_ROWS_WAVES_PER_CU', 'default'), statistics.median(ts), min(ts), max(ts), cs['sum']), flush=True)
""" % ROOT
for w in (None, "8", "12", "16", "32", "128"):
    env = dict(os.environ)
    if w:
        env["PDH_ROWS_WAVES_PER_CU"] = w
    subprocess.run([sys.executable, "-c", CODE], env=env, check=False)
