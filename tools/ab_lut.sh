#!/bin/bash
# developer aid (GPU box): what the bank conflicts of the backward chain's mask-table reads cost -- the product against a build whose table
# reads are conflict-free by construction (every lane of a 16-lane group its own bank quad: wrong masks, timing only), fine-pass shape
python -c "import torch" > /dev/null 2>&1
for rep in 1 2 3; do
  echo -n "product "; MODES=h,h WHAT=chain REPS=10 python tools/bench_mlp.py 2>/dev/null
  echo -n "LUTFREE "; LUSH_SO=build/abl_LUTFREE.so MODES=h,h WHAT=chain REPS=10 python tools/bench_mlp.py 2>/dev/null
done
