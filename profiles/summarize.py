#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>_*.{csv,md} and pmc_traffic.json.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section), so fetch bytes are doubled."""
import csv, glob, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
bench = {}
try:
    bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench json", e)
_pf, _pb = str(bench.get("config", {}).get("planes_fwd", "?")), str(bench.get("config", {}).get("planes_bwd", "?"))
planes = ("h" if _pf.startswith("fp16") else _pf.replace("bf16x", "")) + "," + ("h" if _pb.startswith("fp16") else _pb.replace("bf16x", ""))
# round 3: bench.py runs the timed steps twice (the timed region, then once more with HIP events around each kernel group)
steps = bench.get("steps", 4) * (2 if "kernel_timing" in bench else 1) + bench.get("warmup", 2)
steps += (bench.get("sustained") or {}).get("steps", 0)      # round 5: the >= 3 s of steps behind the timed region (collect.sh passes --sustained 0)

# (checked in this order: the round-5 march runs the wide forward twice -- over all the points without a stash, then with the stash
#  on the live points; the last template argument tells them apart)
GROUP = {"mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 0>": "mlp_fwd_all", "mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 1>": "mlp_fwd",
         "mlp_fwd_kernel<lush::NetT<256": "mlp_fwd", "mlp_bwd_kernel<lush::NetT<256": "mlp_bwd_chain",
         "mlp_chain_fwd_kernel<lush::NetT<256": "mlp_fwd", "mlp_chain_bwd_kernel<lush::NetT<256": "mlp_bwd_chain",
         "mlp_chain_fwd_half_kernel<lush::NetT<256": "mlp_fwd", "mlp_chain_bwd_half_kernel<lush::NetT<256": "mlp_bwd_chain",
         "mlp_wide_fwd_kernel<lush::NetT<256": "mlp_fwd", "mlp_wide_bwd_kernel<lush::NetT<256": "mlp_bwd_chain",
         "dw_group_kernel": "mlp_bwd_weights", "feat_factor_kernel": "mlp_bwd_weights",
         "dw_gemm_kernel": "mlp_bwd_weights", "head_dw_kernel": "mlp_bwd_weights"}


def group_of(name):
    for k, g in GROUP.items():
        if k in name:
            return g
    return None


stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
lines = []
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    out = os.path.join(root, "profiles", f"{tag}_bench_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}',
                        f'{float(r["AverageNs"]) / 1e6:.4f}', r["Percentage"]])
    lines.append(f"## rocprofv3 --kernel-trace --stats of the default `bench.py` command (planes {planes}, {steps} steps incl. warm-up and "
                 f"the kernel-group timing pass; {bench.get('value', '?')} rays/s, {bench.get('ms_per_step', '?')} ms/step inside the profiler)\n")
    lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
    for r in rows[:14]:
        lines.append(f'| `{r["Name"][:70]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e6:.3f} | '
                     f'{float(r["TotalDurationNs"]) / 1e6:.1f} | {r["Percentage"]} |')
    lines.append("\n(The `__amd_rocclr_fillBufferAligned` / `copyBuffer` rows belong to the kernel-group timing pass: its steps run the march "
                 "kernel group by kernel group from Python, with torch allocations in between.  The launches of a step of the TIMED "
                 f"region are listed in `profiles/{tag}_step_trace.txt`.)")
    if bench.get("kernels"):
        lines.append("\nHIP-event timing inside the same run (bench.py `kernels`): " +
                     ", ".join(f'{k}: {v["avg_ms"]} ms/launch-group' for k, v in bench["kernels"].items()))

traffic = {}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(os.path.join(src, f"pmc_{c}", "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        g = group_of(r["Kernel_Name"])
        if g:
            agg[g] += float(r["Counter_Value"]) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
    per[c] = agg
if per:
    lines.append(f"\n## HBM traffic from PMC counters (separate passes; FETCH_SIZE x2 gfx950 correction)\n")
    lines.append("| kernel group | fetch GB/step | write GB/step | total GB/step | algorithmic GB/step |\n|---|---|---|---|---|")
    evals = bench.get("config", {}).get("mlp_evals_per_step", 3932160)
    _n = lambda x: 1 if x == "h" else int(x)
    pf, pb = (_n(planes.split(",")[0]), _n(planes.split(",")[1])) if "?" not in planes else (2, 1)
    # bytes per MLP evaluation and plane (bench.py BYTES_X_STASH / BYTES_DZ_STASH): 1 and 2 planes keep neither
    # the feature activations nor their gradients
    ree = planes == "h,h" or planes == "h,1"      # one fp16 plane, product kernels: 32 bytes per point instead of the 256-byte gamma row
    xs = lambda n: (32 if ree else 256) + 2 * (8 * 256 + (256 if n >= 3 else 0) + 128)
    zs = lambda n: 2 * (8 * 256 + (256 if n >= 3 else 0) + 128 + (8 if n == 1 else 0))
    sp = min(pf, pb)
    alg = {"mlp_fwd": evals * (sp * xs(sp) + 16), "mlp_bwd_chain": evals * (pb * zs(pb) + 336),
           "mlp_bwd_weights": evals * pb * (xs(pb) + zs(pb))}
    lps = {k: v["launches_per_step"] for k, v in bench.get("kernels", {}).items()}
    names = ("mlp_fwd", "mlp_bwd_chain", "mlp_bwd_weights")
    if "mlp_fwd_all" in bench.get("kernels", {}):
        # live-point march: the stash-side kernels run on the live points of each step only -- the algorithmic bytes are the bench
        # line's own (bytes per point of this design x the points each launch processed, from its kernel-group pass)
        names = ("mlp_fwd_all",) + names
        alg = {k: v["hbm_gbs_algorithmic"] * 1e9 * v["avg_ms"] * 1e-3 * v["launches_per_step"] for k, v in bench["kernels"].items()}
    for g in names:
        fe, wr = per.get("FETCH_SIZE", {}).get(g, 0.0) / steps, per.get("WRITE_SIZE", {}).get(g, 0.0) / steps
        lines.append(f"| {g} | {fe / 1e9:.2f} | {wr / 1e9:.2f} | {(fe + wr) / 1e9:.2f} | {alg[g] / 1e9:.2f} |")
        traffic[f"{g}:{planes}"] = {"hbm_bytes_per_launch_group": round((fe + wr) / max(lps.get(g, 2.0), 1e-9)),
                                    "hbm_bytes_per_step": round(fe + wr), "algorithmic_bytes_per_step": round(alg[g]),
                                    "source": f"profiles/{tag}_summary.md"}
    if bench.get("live_points"):
        lines.append(f"\n(Live-point march: {bench['live_points']['share']:.3f} of the points were live in the timed steps; the counters are totals over "
                     f"all {steps} steps of the process, the algorithmic column is the kernel-group pass's points.)")
    json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
if bench:
    json.dump(bench, open(os.path.join(root, "profiles", f"{tag}_bench_line.json"), "w"), indent=1)
open(os.path.join(root, "profiles", f"{tag}_summary.md"), "w").write("# Profile summary " + tag + "\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
