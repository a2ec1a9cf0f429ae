#!/bin/bash
# developer aid (GPU box): same-box A/B of two stream-policy switches (round 5), alternating processes
#   -DLUSH_DW_NT        nt policy on the weight-gradient kernel's LDS-DMA stream
#   -DLUSH_PLAIN_STASH  cached instead of nt stash stores in the forward / chain
python -c "import torch" > /dev/null 2>&1
for rep in 1 2; do
  for so in "" build/dwnt.so; do
    echo "== weights so=${so:-product}"; LUSH_SO=$so MODES=h,h WHAT=weights REPS=8 python tools/bench_mlp.py 2>/dev/null
  done
done
for rep in 1 2; do
  for so in "" build/plainst.so; do
    echo "== fwd,chain,weights so=${so:-product}"; LUSH_SO=$so MODES=h,h WHAT=fwd,chain,weights REPS=8 python tools/bench_mlp.py 2>/dev/null
  done
done
for v in 0 17 0 17; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-kernel-pass --sustained 0 --also "" --extra C1 --variant $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 variant', $v, d['extra_configs']['C1']['ms_per_step'], d['extra_configs']['C1'].get('ms_per_step_eager'))"
done
