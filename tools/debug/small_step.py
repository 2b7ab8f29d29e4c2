"""Step time of a launch-bound problem (BASELINE configs[1]: 2-D, 128^2 cells, 4096 polytopes, p = 2) without per-kernel events:
what the hipGraph replay of pdh_assemble_device buys.  PDH_LIB selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polydeal_amd as pa
import bench

for basis in ("dgp", "dgq"):
    grid, ah, fe = bench.build_handler(pa, 2, 128, 2, basis, 2, 3)
    flat = ah.flatten(pa.SipVariant.poisson_example(fe), True, False)
    ctx = pa.Context(0)
    ctx.set_problem(flat)
    for _ in range(20):
        ctx.assemble_device()
    ctx.synchronize()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(500):
            ctx.assemble_device()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t) / 500)
    print("%s: %.2f us per assembly, %.3e DoF/s (%s)" % (basis, best * 1e6, ah.n_dofs / best, ctx.algorithm_in_use()))
    ctx.close()
