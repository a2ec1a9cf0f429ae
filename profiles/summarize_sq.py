#!/usr/bin/env python3
"""Condense gpurun_out/sq_<tag>/p*/ (rocprofv3 --pmc CSVs) into profiles/<tag>_sq_counters.md: per kernel, per-launch
averages of the SQ counters and the derived shares (gfx950: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles; MI355X_MICROARCH.md, cycle constants)."""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"sq_{tag}")
KEEP = {"mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 1>": "mlp_wide_fwd_kernel (one fp16 plane, stash on, 64 points per wave, 1 workgroup per CU)",
        "mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 0>": "mlp_wide_fwd_kernel (one fp16 plane, no stash)",
        "mlp_chain_fwd_half_kernel<lush::NetT<256": "mlp_chain_fwd_half_kernel (one fp16 plane, stash on, 2 workgroups per CU)",
        "mlp_chain_fwd_kernel<lush::NetT<256": "mlp_chain_fwd_kernel",
        "mlp_chain_bwd_kernel<lush::NetT<256": "mlp_chain_bwd_kernel (one loss-scaled fp16 plane)",
        "mlp_chain_bwd_half_kernel<lush::NetT<256": "mlp_chain_bwd_half_kernel (one loss-scaled fp16 plane, 2 workgroups per CU)",
        "mlp_wide_bwd_kernel<lush::NetT<256": "mlp_wide_bwd_kernel (one loss-scaled fp16 plane, 64 points per wave, 1 workgroup per CU)",
        "dw_group_kernel": "dw_group_kernel (the weight-gradient GEMMs of the fine pass in one launch)"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
newest = {}
for f in glob.glob(os.path.join(src, "p*", "*", "*_counter_collection.csv")):      # (a merged scratch directory may hold earlier collections)
    d = os.path.relpath(f, src).split(os.sep)[0]
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        for k, name in KEEP.items():
            if k in r["Kernel_Name"]:
                agg[name][(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
lines = [f"# SQ counters of the MLP kernel groups ({tag})\n",
         "Command: `rocprofv3 --pmc <set> -- python3 tools/bench_mlp.py` with MODES=h,h WHAT=fwd,chain,weights (fine-pass shape: "
         "20 480 rays x 128 samples; two separate passes of 8 SQ counters, no trace domains; `profiles/collect_sq.sh`).  "
         "Per-launch averages.\n"]
for name, d in agg.items():
    per = collections.defaultdict(list)
    for (c, disp), v in d.items():
        per[c].append(sum(v))
    avg = {c: sum(v) / len(v) for c, v in per.items()}
    lines.append(f"\n## {name}\n\n| counter | per launch |\n|---|---|")
    for c in sorted(avg):
        lines.append(f"| {c} | {avg[c]:.4g} |")
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        cyc = wc * 4
        notes = [f"wave cycles {cyc:.3g}"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg:
            # SQ_BUSY_CYCLES sums over SEs/XCDs in quad-cycles of "any wave active"; the per-SIMD view is wave cycles / waves per SIMD
            pass
        if "SQ_INSTS_MFMA" in avg:
            mf = avg["SQ_INSTS_MFMA"]
            notes.append(f"MFMA instructions {mf:.3g} = {32 * mf:.3g} matrix-pipe cycles; "
                         f"matrix pipe busy / wave cycles = {32 * mf / cyc:.2f} (x waves per SIMD for the per-SIMD share)")
            tot = sum(avg.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
            if tot:
                notes.append(f"non-MFMA instructions per MFMA {(tot - mf) / mf:.1f}")
        for k, label in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "parked in s_waitcnt / s_barrier")):
            if k in avg:
                notes.append(f"{label} {avg[k] / wc:.0%}")
        if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
            notes.append(f"LDS bank-conflict cycles / LDS active cycles {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']:.1%}")
        lines.append("\nDerived: " + "; ".join(notes) + ".")
open(os.path.join(root, "profiles", f"{tag}_sq_counters.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
