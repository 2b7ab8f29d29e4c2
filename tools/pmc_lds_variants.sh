#!/bin/bash
# LDS counters of the row kernel with one phase removed at a time (libraries built with -DPDHR_EXP=1..4): attributes
# bank conflicts / LDS activity to phases by difference.  Usage (GPU box): tools/pmc_lds_variants.sh
export TMPDIR=/tmp
for v in 0 1 2 3 4; do
  OUT=gpurun_out/pmc_ldsv$v
  mkdir -p $OUT
  if [ $v -eq 0 ]; then unset PDH_LIB; else export PDH_LIB=$PWD/build/rv$v/libpolydeal_hip.so; fi
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/a -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-aux-kernels > $OUT/a.json 2> $OUT/a.err
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/a/p_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if "k_rows" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant $v", {k: "%.3g" % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
done
