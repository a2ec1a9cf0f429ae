#!/bin/bash
# Developer helper for one gpurun call: run steps in order, stop at the first step that was killed (timeout / signal).
# usage: bash tests/gpu_session.sh <tag> <step> [<step> ...]   where a step is a quoted shell command
TAG=$1; shift
mkdir -p gpurun_out
i=0
for step in "$@"; do
  i=$((i+1))
  echo "=== step $i: $step" | tee -a gpurun_out/${TAG}_session.log
  bash -c "$step" >> gpurun_out/${TAG}_step$i.log 2>&1
  rc=$?
  echo "=== step $i rc=$rc" | tee -a gpurun_out/${TAG}_session.log
  tail -n 6 gpurun_out/${TAG}_step$i.log
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step killed: stopping"; exit $rc; fi
done
exit 0
