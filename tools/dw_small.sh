#!/bin/bash
# developer aid (GPU box): lush_mlp_bwd_weights of the 8x256 net over point counts, for the product library and any variant
# libraries named on the command line (tools/build_variant.py):  bash tools/dw_small.sh [build/x.so ...]
for so in "" "$@"; do
  echo "== ${so:-product}"
  for rs in "64 64" "512 64" "4096 64" "8192 64" "20480 64" "20480 128"; do set -- $rs; LUSH_SO=$so R=$1 S=$2 MODES=h,h WHAT=weights REPS=5 python tools/bench_mlp.py 2>&1 | tail -1; done
done
