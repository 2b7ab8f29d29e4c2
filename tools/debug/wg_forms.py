#!/usr/bin/env python3
"""FE_DGQ(3) on the bench mesh: the workgroup term kernel (cells / sub-faces merged into composite rules, or as given) against pdh_rows.h;
each form in its own child process, several contexts each (the spread between contexts is the allocation placement).  (profiles/
r04_wg_forms.txt was taken while an "entry form" of phase A still existed behind PDH_TERMS_WG_ENTRY.)
usage: wg_forms.py [cells=64] [contexts=3]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, statistics
sys.path.insert(0, %r)
import bench, polydeal_amd as pa
cells, nctx = int(sys.argv[1]), int(sys.argv[2])
grid, ah, fe = bench.build_handler(pa, 3, cells, 2, "dgq", 3, 4)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
ctxs = []
for _ in range(nctx):
    c = pa.Context(0); c.set_overlap(False); c.set_problem(flat); ctxs.append(c)
kern = ctxs[0].rows_kernel_in_use()
for c in ctxs:
    for _ in range(3): c.assemble_device()
    c.synchronize()
ts = []
for rnd in range(4):
    for c in ctxs:
        c.set_profiling(True)
        for _ in range(4): c.assemble_device()
        (k0, k1), _ = c.kernel_times_ms(); c.set_profiling(False)
        ts.append(k0 + k1)
print("%%-34s %%-7s median %%.3f ms  min %%.3f  max %%.3f  sum %%.10e" %% (sys.argv[3], kern, statistics.median(ts), min(ts), max(ts), ctxs[0].checksum()["sum"]), flush=True)
''' % ROOT
cells = sys.argv[1] if len(sys.argv) > 1 else "64"
nctx = sys.argv[2] if len(sys.argv) > 2 else "3"
forms = [("pdh_rows.h", dict(PDH_TERMS_DGQ3="0")),
         ("terms_wg merged", dict(PDH_TERMS_DGQ3="1", PDH_TERMS_MERGE="1")),
         ("terms_wg as given", dict(PDH_TERMS_DGQ3="1", PDH_TERMS_MERGE="0")),
         ("terms_wg merged (again)", dict(PDH_TERMS_DGQ3="1", PDH_TERMS_MERGE="1")),
         ("pdh_rows.h (again)", dict(PDH_TERMS_DGQ3="0"))]
for name, env in forms:
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, "-c", CHILD, cells, nctx, name], env=e, check=False)
