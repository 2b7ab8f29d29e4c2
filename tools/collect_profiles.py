#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/ into profiles/ and refresh
profiles/traffic.json.  Usage: tools/collect_profiles.py TAG  (reads gpurun_out/prof_TAG/)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % tag)
line = open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1]
open("profiles/%s_bench_under_rocprof.json" % tag, "w").write(line + "\n")
bench = json.loads(line)
out, summ = {}, []
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(os.path.join(src, "pmc_%s" % name, "bench_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        summ.append(dict(kernel=k, counter=c, dispatches=len(v), mean_KB=sum(v) / len(v), min_KB=min(v), max_KB=max(v)))
        out[(k.split("<")[0].split("::")[-1], c)] = sum(v) / len(v)
json.dump(summ, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)


def tr(k):
    return 2 * out[(k, "FETCH_SIZE")] * 1024 + out[(k, "WRITE_SIZE")] * 1024


tpath = "profiles/traffic.json"
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
wl = "3D cells=64 block=2 dgq p=3"
entry = {}  # only what THIS run measured: byte counts of older kernels must not survive under a new library version
for kname in sorted({k for (k, c) in out}):
    if (kname, "FETCH_SIZE") in out and (kname, "WRITE_SIZE") in out and kname.startswith("k_") and "(" not in kname:
        entry[kname + "_bytes"] = tr(kname)
entry["round"] = tag
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polydeal_amd  # noqa: E402
entry["lib_version"] = polydeal_amd.load_library().pdh_version().decode()
entry["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/%s_pmc_summary.json); bytes = "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of coalesced streaming reads "
                 "(MI355X_MICROARCH.md, HBM); checked: 2*FETCH matches the algorithmic read bytes of k_diag within a few %%" % tag)
t[wl] = entry
json.dump(t, open(tpath, "w"), indent=1)
print(open("profiles/%s_kernel_stats.csv" % tag).read())
print("bench under rocprof: value %.4g ms/step %.3f dominant kernel %s %.3f ms" % (
    bench["value"], bench["ms_per_step"], bench["roofline"]["kernel"], bench["roofline"]["kernel_ms"]))
print("cpu:", bench["cpu_baseline"])
print("traffic:", {k: v for k, v in entry.items() if k.endswith("_bytes")})
print("alg bytes dominant kernel:", bench["roofline"]["algorithmic_bytes_per_launch"])
