#!/bin/bash
# developer aid (GPU box): what dropping h_0 from the stash could gain at most (timing ablation -DLUSH_ABL_H0: the forward's h_0 rows
# stay on chip, the weight-gradient job of layer 1 streams no X), alternating processes
python -c "import torch" > /dev/null 2>&1
for rep in 1 2 3; do
  for so in "" build/ablh0.so; do
    echo -n "so=${so:-product} "; LUSH_SO=$so MODES=h,h WHAT=fwd,weights REPS=10 python tools/bench_mlp.py 2>/dev/null
  done
done
