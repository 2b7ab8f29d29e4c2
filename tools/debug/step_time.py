"""Whole-step wall time of the default bench workload without per-kernel events (experiments)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polydeal_amd as pa
import bench

class A: pass
a = A(); a.dim, a.cells, a.block, a.degree = 3, 64, 2, 3
basis = sys.argv[1] if len(sys.argv) > 1 else "dgq"
grid, ah, fe = bench.build_handler(pa, a.dim, a.cells, a.block, basis, a.degree, a.degree + 1)
flat = ah.flatten(pa.SipVariant.poisson_example(fe), diag_first=os.environ.get("DIAG_FIRST", "1") == "1", with_colind=False)
ctx = pa.Context(0)
ctx.set_problem(flat)
for _ in range(3):
    ctx.assemble_device()
ctx.synchronize()
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(20):
        ctx.assemble_device()
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t) / 20)
ctx.set_profiling(True)
for _ in range(10):
    ctx.assemble_device()
print("LDS pad %s: kernel ms" % os.environ.get("PDH_EXP_LDS_PAD", "0"), ctx.kernel_times_ms(), ctx.stats()["lds_bytes_diag"])
ctx.set_profiling(False)
print("DIAG_FIRST=%s PDH_TWO_STREAMS=%s %s: %.3f ms/step" % (os.environ.get("DIAG_FIRST", "1"), os.environ.get("PDH_EXP_TWO_STREAMS", "0"), basis, best * 1e3))
