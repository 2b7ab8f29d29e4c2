#!/bin/bash
# Instruction-cache counters of a short bench run (GPU box).  Usage: tools/pmc_icache.sh TAG [bench args]
TAG=${1:-ic}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_TC_[A-Z_]*\|SQ_BUSY_CU_CYCLES\|SQ_WAVES\b" $OUT/counters.txt | sort -u | tr '\n' ' '
echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
python3 - <<PY
import csv, collections
try:
    rows=list(csv.DictReader(open("$OUT/a/p_counter_collection.csv")))
except Exception as e:
    print("no csv", e); rows=[]
agg=collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; print("%-36s %-28s n=%d mean=%.4g"%(k[0],k[1],len(v),sum(v)/len(v)))
PY
tail -3 $OUT/a.err
