#!/bin/bash
# developer aid (GPU box): kernel times of one bench step's small launches per variant library:  bash tools/noise_dw.sh build/x.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for so in "" "$@"; do
  rm -rf gpurun_out/trace_tmp
  if [ -z "$so" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --also "" --extra "" --no-traffic --no-kernel-pass > /dev/null 2>&1
  else rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --also "" --extra "" --no-traffic --no-kernel-pass --so $so > /dev/null 2>&1; fi
  echo "== ${so:-product}"; python tools/trace_step.py gpurun_out/trace_tmp | grep "everything else\|dw_group_kernel<false"
done
rm -rf gpurun_out/trace_tmp
