#!/bin/bash
# developer aid (GPU box): the grouped weight-gradient launch as the walk (product) against one job per workgroup on cost-proportional
# slices (VARIANT=128 = LUSH_VARIANT_DW_SPLIT, round 5 experiment), fine and coarse pass shapes, alternating processes
python -c "import torch" > /dev/null 2>&1
for S in 128 64; do
  for rep in 1 2 3; do
    for v in 0 128; do
      echo -n "S=$S variant=$v "; S=$S VARIANT=$v MODES=h,h WHAT=weights REPS=10 python tools/bench_mlp.py 2>/dev/null
    done
  done
done
