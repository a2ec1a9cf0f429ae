#!/bin/bash
# developer aid (GPU box): where do the 128-point-tile kernels (variant 17 = FWD_HALF | BWD_HALF) beat the 256-point-tile ones?
# one pass of R rays x 32 samples, forward (stash) / chain / weight gradients, per variant
python -c "import torch" > /dev/null 2>&1
for R in 128 256 512 1024 2048; do
  for v in 0 17; do
    echo -n "R=$R S=32 variant=$v: "; R=$R S=32 VARIANT=$v MODES=h,h WHAT=fwd,chain,weights REPS=20 python tools/bench_mlp.py 2>/dev/null
  done
done
