#!/bin/bash
# LDS / vector-memory pipeline counters of a short bench run (GPU box).  Usage: tools/pmc_mem.sh TAG [bench args]
TAG=${1:-mem}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --kernel-trace --output-format csv -d $OUT/a -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/b -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err
python3 - <<PY
import csv, collections
for part in "ab":
    try:
        rows=list(csv.DictReader(open("$OUT/%s/p_counter_collection.csv"%part)))
    except Exception as e:
        print("no csv", part, e); continue
    agg=collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        if "rocclr" in k[0]: continue
        v=agg[k]; print("%-36s %-30s n=%d mean=%.4g"%(k[0],k[1],len(v),sum(v)/len(v)))
PY
