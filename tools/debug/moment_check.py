"""Moment form vs direct form vs oracle on small 3-D cases, then timing on the default bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polydeal_amd as pa
from oracle import polydeal_oracle as po
from flatten_oracle import flatten
from test_gpu_parity import build

def run(kw, alg):
    prob = pa.Problem(**kw)
    ctx = pa.Context(0)
    ctx.set_problem(prob)
    ctx.set_algorithm(alg)
    v = ctx.assemble()
    used = ctx.algorithm_in_use()
    ctx.close()
    return v, used

for (fe_cls, p, lg, b, dist, varname) in [(po.FE_DGQ, 3, 2, 2, 0.0, "poisson"), (po.FE_DGQ, 3, 2, 2, 0.15, "dr"),
                                          (po.FE_DGQ, 2, 2, 2, 0.1, "dr"), (po.FE_DGQ, 1, 2, 2, 0.1, "poisson"),
                                          (po.FE_AggloDGP, 3, 2, 2, 0.1, "poisson"), (po.FE_AggloDGP, 2, 2, 1, 0.0, "adm"),
                                          (po.FE_DGQ, 3, 1, 2, 0.0, "poisson")]:
    fe = fe_cls(3, p)
    ah = build(3, lg, b, fe, p + 1, distort=dist)
    var = {"poisson": lambda: po.variant_poisson_example(fe), "dr": lambda: po.variant_diffusion_reaction(fe),
           "adm": po.variant_assemble_dg_matrix}[varname]()
    kw = flatten(ah, var)
    ref = po.assemble_csr(ah, var)[2]
    vd, _ = run(kw, "direct")
    vm, used = run(kw, "moment")
    sc = np.max(np.abs(ref))
    print("%s(%d) lg=%d b=%d dist=%.2f %s: direct %.2e  moment(%s) %.2e" % (fe.name, p, lg, b, dist, varname,
          np.max(np.abs(vd - ref)) / sc, used, np.max(np.abs(vm - ref)) / sc), flush=True)

if len(sys.argv) > 1:
    import bench
    for basis in ("dgq", "dgp"):
        grid, ah, fe = bench.build_handler(pa, 3, 64, 2, basis, 3, 4)
        flat = ah.flatten(pa.SipVariant.poisson_example(fe), diag_first=True, with_colind=False)
        ctx = pa.Context(0)
        ctx.set_problem(flat)
        ctx.set_overlap(False)
        for alg in ("direct", "moment"):
            ctx.set_algorithm(alg)
            for _ in range(2):
                ctx.assemble_device()
            ctx.synchronize()
            ctx.set_profiling(True)
            t = time.perf_counter()
            for _ in range(10):
                ctx.assemble_device()
            ctx.synchronize()
            dt = (time.perf_counter() - t) / 10
            print(basis, alg, "%.3f ms/step" % (dt * 1e3), ctx.kernel_times_ms(), flush=True)
            ctx.set_profiling(False)
        ctx.close()
