#!/bin/bash
# developer aid (GPU box): engine clock and package power while one kernel group of the fine pass runs in a loop
#   bash tools/power_probe.sh [groups...]            (groups: fwd, fwd_nostash, chain, weights)
python -c "import torch" > /dev/null 2>&1          # (the first import on a fresh box takes a minute)
for what in ${@:-fwd fwd_nostash chain weights}; do
  MODES=h,h WHAT=$what REPS=3000 python tools/bench_mlp.py > /dev/null 2>&1 &
  pid=$!
  echo "== $what"
  for i in $(seq 1 14); do
    sleep 1
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' '; echo
  done
  wait $pid
done
