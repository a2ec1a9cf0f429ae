#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh r04
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (headline mode only)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command (MI355X_MICROARCH.md: TCC slots)
# Raw CSVs go to gpurun_out/ (scratch); profiles/summarize.py condenses them into profiles/<tag>_* and pmc_traffic.json.
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --also= --extra= --no-traffic --sustained 0 --no-dense"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$c.json 2> $OUT/pmc_$c.err
done
cd $ROOT && python3 profiles/summarize.py $TAG && python3 tools/trace_step.py $OUT/trace --step 3 > profiles/${TAG}_step_trace.txt
find $OUT -name "*.csv" -size +30M -delete
