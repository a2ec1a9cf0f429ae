#!/bin/bash
# developer aid (GPU box): forward / chain of the fine pass, product against variant libraries:  bash tools/ab_fwd.sh build/x.so ...
for so in "" "$@"; do
  echo "== ${so:-product}"
  LUSH_SO=$so MODES=h,h WHAT=fwd,chain REPS=5 python tools/bench_mlp.py 2>&1 | tail -1
done
