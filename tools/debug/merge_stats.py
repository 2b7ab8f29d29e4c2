import os, sys
sys.path.insert(0, "/root/repo")
import bench, polydeal_amd as pa
for cs in ("dgp3", "dgq2"):
    basis, p = cs[:3], int(cs[3])
    grid, ah, fe = bench.build_handler(pa, 3, 32, 2, basis, p, p + 1, grown=True)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    for m in ("2", "0"):
        os.environ["PDH_TERMS_MERGE"] = m
        ctx = pa.Context(0); ctx.set_problem(flat); print(cs, "merge", m, ctx.rows_kernel_in_use(), ctx.terms_merge_stats(), flush=True); ctx.close()
