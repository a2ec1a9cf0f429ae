#!/bin/bash
# developer aid (GPU box): what the weight-gradient launch's fp32 atomics cost at the point counts a live-point launch has --
# product against a -DLUSH_ABL_NOFLUSH build (wrong results, timing only), dense launches of R rays x 128 samples, alternating processes
python tools/build_variant.py --out build/noflush.so --flags=-DLUSH_ABL_NOFLUSH > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for R in 2048 4096 8192 12288 20480; do
  for rep in 1 2; do
    echo -n "R=$R product  "; R=$R S=128 MODES=h,h WHAT=weights REPS=10 python tools/bench_mlp.py 2>/dev/null
    echo -n "R=$R noflush  "; LUSH_SO=build/noflush.so R=$R S=128 MODES=h,h WHAT=weights REPS=10 python tools/bench_mlp.py 2>/dev/null
  done
done
