#!/bin/bash
# developer aid (GPU box): SQ counters of the two-wave prototype (tools/micro/$BIN.hip), two --pmc passes, no trace domains
set -u
BIN=${1:-two_wave_proto}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_proto; mkdir -p $OUT
rm -rf $OUT/p1 $OUT/p2
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
timeout -k 10 200 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -- $ROOT/build/$BIN > $OUT/p1.log 2> $OUT/p1.err
timeout -k 10 200 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -- $ROOT/build/$BIN > $OUT/p2.log 2> $OUT/p2.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/sq_proto/{p}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            gs = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
            if gs >= 256 * 512 or gs == 0:      # the timed launches (the 2-workgroup correctness launch is skipped)
                out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in out.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print("##", k.split("(")[0])
    print("  " + ", ".join(f"{n} {v:.4g}" for n, v in sorted(m.items())))
    if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"]:
        non = sum(m.get(n, 0) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) - m["SQ_INSTS_MFMA"]
        print(f"  Derived: non-MFMA instructions per MFMA {non / m['SQ_INSTS_MFMA']:.1f}; matrix pipe busy / wave cycles = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_WAVE_CYCLES']):.2f} per wave"
              f" (x2 waves per SIMD); issuing {100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f}%, issue-stalled {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f}%, "
              f"parked in s_waitcnt / s_barrier {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.0f}%; LDS bank conflicts {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.1f}% of LDS active cycles")
PY
