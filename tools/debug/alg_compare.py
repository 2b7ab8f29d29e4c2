"""direct vs moment form on a bench workload: python tools/debug/alg_compare.py CELLS DEGREE FE VARIANT"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import polydeal_amd as pa
import bench
cells, degree, fe_name, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
grid, ah, fe = bench.build_handler(pa, 3, cells, 2, fe_name, degree, degree + 1)
class A: pass
flat = ah.flatten(bench.make_variant(pa, variant, fe), diag_first=True, with_colind=False)
ctx = pa.Context(0)
ctx.set_problem(flat)
for alg in ("direct", "moment"):
    ctx.set_algorithm(alg)
    for _ in range(2):
        ctx.assemble_device()
    ctx.synchronize()
    ctx.set_profiling(True)
    t = time.perf_counter()
    for _ in range(5):
        ctx.assemble_device()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(cells, degree, fe_name, variant, alg, "%.3f ms/step" % (dt * 1e3), ctx.kernel_times_ms()[0], flush=True)
    ctx.set_profiling(False)
ctx.close()
