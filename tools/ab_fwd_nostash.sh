#!/bin/bash
# developer aid (GPU box): where the no-stash forward's time goes -- the product against timing-only ablation builds (wrong results):
# no positional encoding, no position barrier, no conversion (the MFMA + fragment / DMA skeleton), no weight DMAs; fine-pass shape
python -c "import torch" > /dev/null 2>&1
for rep in 1 2; do
  echo -n "product "; MODES=h,h WHAT=fwd_nostash REPS=10 python tools/bench_mlp.py 2>/dev/null
  for f in NOPE NOBAR NOCONV NODMA; do
    echo -n "$f "; LUSH_SO=build/abl_$f.so MODES=h,h WHAT=fwd_nostash REPS=10 python tools/bench_mlp.py 2>/dev/null
  done
done
