#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench command, then the two
# PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots).
# Outputs under gpurun_out/prof_$1/ ; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 10 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
find $OUT -name "*.csv" | head -20
