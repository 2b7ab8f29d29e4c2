#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels of a short bench run (GPU box).  Usage: tools/pmc_hbm.sh TAG [bench args]
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of coalesced streaming reads, MI355X_MICROARCH.md)
TAG=${1:-hbm}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/f.json 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/w.json 2> $OUT/w.err
python3 - <<PY
import csv, collections, json
agg=collections.defaultdict(list)
for part in "fw":
    for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv"%part)):
        agg[(r["Kernel_Name"].replace("void ","").split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
names=sorted({k for k,_ in agg})
for k in names:
    f=agg.get((k,"FETCH_SIZE")); w=agg.get((k,"WRITE_SIZE"))
    if f and w and "rocclr" not in k and "at::" not in k:
        F=sum(f)/len(f)*1024; W=sum(w)/len(w)*1024
        print("%-60s n=%d read %.4g GB (2 x FETCH_SIZE) written %.4g GB total %.4g GB" % (k, len(w), 2*F*1e-9, W*1e-9, (2*F+W)*1e-9))
d=json.loads(open("$OUT/w.json").read().strip().splitlines()[-1])
print("algorithmic bytes of the dominant kernel: %.4g GB (%s)" % (d["roofline"]["algorithmic_bytes_per_launch"]*1e-9, json.dumps(d["roofline"].get("algorithmic_bytes_parts"))))
PY
