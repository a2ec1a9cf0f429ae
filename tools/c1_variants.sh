#!/bin/bash
# developer aid (GPU box): BASELINE config 1 (8 192 evaluations per step, replayed as one HIP graph) per kernel-variant word
for v in 0 17 6 ; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-kernel-pass --also "" --extra C1 --variant $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', $v, d['extra_configs']['C1']['ms_per_step'], d['extra_configs']['C1'].get('ms_per_step_eager'))"
done
