#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  [WHAT=fwd,fwd_nostash,chain,weights] [VARIANT=0] bash profiles/collect_sq.sh r03
# SQ counters of the three MLP kernel groups at the fine-pass shape (20 480 rays x 128 samples, headline mode h,h):
# two separate rocprofv3 --pmc passes (8 SQ slots each), no trace domains.  Raw CSVs go to gpurun_out/ (scratch);
# profiles/summarize_sq.py condenses them into profiles/<tag>_sq_counters.md.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODES="${MODES:-h,h}" WHAT="${WHAT:-fwd,fwd_nostash,chain,weights}" REPS=2     # (bench_mlp.py: 1 untimed + REPS timed launches per group)
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/bench_mlp.py > $OUT/p1.log 2> $OUT/p1.err
timeout -k 10 300 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/bench_mlp.py > $OUT/p2.log 2> $OUT/p2.err
tail -n 2 $OUT/p1.log; tail -n 2 $OUT/p2.log
