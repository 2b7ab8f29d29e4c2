#!/bin/bash
# SQ counter pass over a short bench run (GPU box).  Usage: tools/pmc_sq.sh TAG [bench args]
TAG=${1:-sq}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/a -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/b -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err
python3 - <<PY
import csv, collections
for part in "ab":
    try:
        rows=list(csv.DictReader(open("$OUT/%s/p_counter_collection.csv"%part)))
    except Exception as e:
        print("no csv", part, e); continue
    agg=collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        v=agg[k]; print("%s | %-28s n=%d mean=%.4g"%(k[0].replace("void ","").split("(")[0],k[1],len(v),sum(v)/len(v)))  # full kernel names: the instantiations differ
PY
tail -2 $OUT/a.err
