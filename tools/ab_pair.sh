#!/bin/bash
# developer aid (GPU box): the row-split wave-pair prototype (tools/micro/pair_proto.hip) beside the product's inference forward, same box
python -c "import torch" > /dev/null 2>&1
for rep in 1 2; do
  echo -n "product fwd_nostash "; MODES=h,h WHAT=fwd_nostash REPS=10 python tools/bench_mlp.py 2>/dev/null
  for b in "$@"; do echo "== $b"; timeout -k 10 120 ./build/$b || exit 1; done
done
