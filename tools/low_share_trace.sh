#!/bin/bash
# Run ON THE GPU BOX: kernel trace of tools/live_steps.py long enough for the synthetic fit to collapse (live share ~0.02),
# then the launches of one late step -- what the live-point backward costs when almost nothing is live.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/low_share
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-420} POLICY=live timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/live_steps.py > $OUT/run.log 2> $OUT/run.err
cd $ROOT && python3 tools/trace_step.py $OUT/trace --step ${STEP:-415} > $OUT/step.txt 2>&1
cat $OUT/step.txt
find $OUT -name "*.csv" -size +30M -delete
