#!/bin/bash
# developer aid (GPU box): weight gradients with the gamma re-encode on all eight waves (product) against four (build/dwnt.so, the
# tree before that change), fine and coarse pass shapes, alternating processes
python -c "import torch" > /dev/null 2>&1
for S in 128 64; do
  for rep in 1 2 3; do
    for so in "" build/dwnt.so; do
      echo -n "S=$S so=${so:-product} "; S=$S LUSH_SO=$so MODES=h,h WHAT=weights REPS=10 python tools/bench_mlp.py 2>/dev/null
    done
  done
done
