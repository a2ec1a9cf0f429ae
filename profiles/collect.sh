#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command (MI355X_MICROARCH.md: TCC slots)
# Raw CSVs go to gpurun_out/ (scratch); profiles/summarize.py condenses them into profiles/.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --variant 32 (LUSH_VARIANT_NO_OVERLAP): kernels one at a time, so that a kernel's duration and traffic are its own -- what
# bench.py's kernel-group timing pass and `roofline` measure.  The default command runs the fine pass's weight gradients beside the
# coarse pass's chain: one more kernel trace of it goes to trace_overlap/.
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --also= --extra= --variant 32"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$c.json 2> $OUT/pmc_$c.err
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --also= --extra= > $OUT/bench_overlap.json 2> $OUT/trace_overlap.err
cd $ROOT && python3 profiles/summarize.py $TAG
