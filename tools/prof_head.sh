#!/bin/bash
# usage: prof_head.sh tag  (env passes through)
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ph_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODES="h,h" WHAT=weights REPS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_mlp.py > $OUT/log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if 'head_dw' in r['Name'] or 'dw_group' in r['Name'] or 'feat_factor' in r['Name']:
        print("$TAG", r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, 'us')
PY
