#!/usr/bin/env python3
"""One source of truth for the numbers quoted in DESIGN.md and README.md.

  python tools/quoted_numbers.py            # rewrites the blocks between <!-- numbers:NAME --> ... <!-- /numbers:NAME -->
  python tools/quoted_numbers.py --check    # exit 1 if a block is out of date (tests/test_cpu_host.py runs this)

Sources, nothing else: the driver's latest BENCH_rNN.json at the repo root (the HEADLINE: what the driver measured on its
own box at the end of the previous round), and this round's committed profile artefacts under profiles/ (rNN_bench_line.json =
the bench line of the profiled run, rNN_bench_kernel_stats.csv, rNN_sq_counters.md, pmc_traffic.json, rNN_step_trace.txt).
Boxes differ by 2-4 % on the same binary, so a figure always names its source."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def latest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, pattern)))
    return fs[-1] if fs else None


def driver_bench():
    """The driver's file the status block quotes: the one of the round BEFORE the newest committed profile set (profiles/rNN_* are
    this round's, so the driver's latest file when they were made is BENCH_r(NN-1).json).  Tied to the profile tag, not to "the
    newest BENCH file in the directory": the driver adds BENCH_rNN.json after the round and the committed text must not go stale
    by that -- but a round that commits its rNN profiles while still quoting an older driver file fails the check (round 5 quoted
    BENCH_r04.json beside r05 profiles)."""
    tag = profile_tag()
    f = None
    if tag:
        cand = os.path.join(ROOT, "BENCH_r%02d.json" % (int(tag[1:]) - 1))
        f = cand if os.path.exists(cand) else None
    f = f or latest("BENCH_r[0-9][0-9].json")
    if not f:
        return None, None
    d = json.load(open(f))
    return os.path.basename(f), d.get("parsed") or d


def profile_tag():
    f = latest("profiles/r[0-9][0-9]_bench_line.json")
    return os.path.basename(f)[:3] if f else None


def blocks():
    out = {}
    name, drv = driver_bench()
    tag = profile_tag()
    line = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_line.json"))) if tag else {}
    # ---- status line (README)
    s = []
    if drv:
        evals = drv["config"].get("mlp_evals_per_step", 3932160)
        frac = 6 * 593408 * evals / (drv["ms_per_step"] * 1e-3) / 1e12 / PEAK_TF        # fwd + dX + dW, algorithmic, of the dense bf16 peak
        cpu = drv.get("cpu_baseline") or {}
        s.append(f"**Headline (the driver's own run, `{name}`): {drv['value']:,.0f} input rays/s, {drv['ms_per_step']:.3f} ms/step at (h,h)** "
                 f"(BASELINE config 2, 1 x MI355X, mode `{drv['config'].get('planes_fwd', '?')}` / `{drv['config'].get('planes_bwd', '?')}`; "
                 f"whole step {frac:.2f} of the dense bf16 MFMA peak counting dead points; dominant kernel `{drv['roofline']['kernel']}` at "
                 f"{drv['roofline']['frac']:.2f} of the {drv['roofline']['bound'].upper()} roof; CPU oracle on the box's {cpu.get('cores', '?')} host threads "
                 f"{cpu.get('value', 0):.1f} rays/s).  The driver's record keeps the contract's keys only; what that figure depends on is in the "
                 "full line of the same command:")
    un = os.path.join(ROOT, "profiles", f"{tag}_bench_unprofiled.json") if tag else None
    if un and os.path.exists(un):
        u = json.load(open(un))
        lp, tl, rf = u.get("live_points") or {}, (u.get("extra_configs") or {}).get("trained_like") or {}, u.get("roofline") or {}
        s.append(f"**This round's un-profiled run of the driver's command (`profiles/{tag}_bench_unprofiled.json`, another box): {u['value']:,.0f} rays/s, "
                 f"{u['ms_per_step']:.3f} ms/step at (h,h) and a live share of {lp.get('share', 0):.2f}** (the timed steps and their draws are "
                 f"deterministic: the same share in every run); **with the backward over all the points -- the step whose cost does not depend on the "
                 f"model's state -- {u.get('value_dense', 0):,.0f} rays/s, {u.get('ms_per_step_dense', 0):.2f} ms**; from a density field shaped like a "
                 f"trained scene's (`extra_configs.trained_like`, live share {(tl.get('live_points') or {}).get('share', 0):.2f}) "
                 f"**{tl.get('value', 0):,.0f} rays/s, {tl.get('ms_per_step', 0):.2f} ms**; strict fp32-equivalent mode (2,2) {u.get('value_strict_2_2', 0):,.0f} rays/s; "
                 f"dominant kernel `{rf.get('kernel')}` at {rf.get('frac', 0):.2f} of the 2.5-PF datasheet peak = {rf.get('frac_of_sustained_peak') or 0:.2f} of the "
                 f"{rf.get('sustained_peak') or 0:,.0f} TFLOP/s a bare MFMA loop of its shape sustains on this chip (DESIGN.md section 5).")
    if line:
        lp = line.get("live_points")
        s.append(f"This round's profiled run (`profiles/{tag}_bench_line.json`, `--steps 4`, inside `rocprofv3`): {line['value']:,.0f} rays/s, "
                 f"{line['ms_per_step']:.3f} ms/step" + (f" at a live share of {lp['share']:.2f}" if lp else "") + ".")
    out["status"] = "\n".join(s)
    # ---- every configuration of this round's un-profiled line (README)
    if un and os.path.exists(un):
        u = json.load(open(un))
        ex = u.get("extra_configs") or {}
        rows = ["| workload (1 x MI355X, mode (h,h) unless said) | rays/s | ms/step | note |", "|---|---|---|---|"]
        rows.append(f"| BASELINE config 2 (N_rand 4096, 64 + 64, blur kernel on): `value` | {u['value']:,.0f} | {u['ms_per_step']:.3f} | live share {(u.get('live_points') or {}).get('share', 0):.2f} (a freshly initialised model) |")
        if u.get("value_dense"):
            rows.append(f"| ... backward over all the points (`value_dense`) | {u['value_dense']:,.0f} | {u['ms_per_step_dense']:.2f} | independent of the model's state |")
        tl = ex.get("trained_like")
        if tl:
            rows.append(f"| ... from a trained-like density field (`extra_configs.trained_like`) | {tl['value']:,.0f} | {tl['ms_per_step']:.3f} | live share {(tl.get('live_points') or {}).get('share', 0):.2f}; {tl.get('backward', '')} |")
        for m, lab in (("2,2", "strict fp32-equivalent mode (2,2)"), ("2,h", "mode (2,h)"), ("2,1", "mode (2,1)")):
            if m in (u.get("modes") or {}):
                rows.append(f"| ... {lab} | {u['modes'][m]['value']:,.0f} | {u['modes'][m]['ms_per_step']:.2f} | |")
        for c, lab in (("C1", "BASELINE config 1 (N_rand 256, 32 + 0, naive), replayed HIP graph"), ("C3", "BASELINE config 3 (N_rand 8192, 64 + 64)"),
                       ("C5", "BASELINE config 5, per GPU (N_rand 16384, 128 + 128)")):
            if c in ex:
                rows.append(f"| {lab} | {ex[c]['value']:,.0f} | {ex[c]['ms_per_step']:.3f} | |")
        if "eval" in ex:
            rows.append(f"| eval path, forward only (640 x 1120 rays per pose) | {ex['eval']['value']:,.0f} | {ex['eval']['s_per_pose'] * 1e3:.1f} per pose | |")
        cb = u.get("cpu_baseline") or {}
        if cb:
            rows.append(f"| CPU oracle on the box's {cb.get('cores')} host threads (N_rand 512, kernel on) | {cb.get('value', 0):.1f} | {cb.get('s_per_step', 0) * 1e3:,.0f} | `cpu_baseline` (kind: port); GPU / CPU = {u.get('gpu_over_cpu', 0):,.0f} |")
        rows.append(f"\n(`profiles/{tag}_bench_unprofiled.json`: one `python bench.py --steps 20 --warmup 5` on one box; boxes differ by 2 - 4 %.)")
        out["configs"] = "\n".join(rows)
    # ---- kernel table
    if line:
        rows = ["| group | launches / step | avg launch ms (HIP events) | algorithmic TFLOP/s (of 2.5 PF) | algorithmic HBM GB/s (of 8 TB/s) | PMC HBM GB / step (algorithmic) |",
                "|---|---|---|---|---|---|"]
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))) if os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json")) else {}
        live = "mlp_fwd_all" in line["kernels"]
        label = {"mlp_bwd_weights": "dW (`dw_group_kernel`)" + (", live points" if live else ""),
                 "mlp_fwd": "forward with the stash (`mlp_wide_fwd_kernel<.., 1>`)" + (", live points" if live else ""),
                 "mlp_fwd_all": "forward, all points, no stash (`mlp_wide_fwd_kernel<.., 0>`)",
                 "mlp_bwd_chain": "dX chain (`mlp_wide_bwd_kernel`)" + (", live points" if live else "")}
        for g in sorted(line["kernels"], key=lambda g: -line["kernels"][g]["ms_per_step"]):
            k = line["kernels"][g]
            t = tr.get(f"{g}:h,h", {})
            rows.append(f"| {label[g]} | {k['launches_per_step']:.0f} | {k['avg_ms']:.3f} | {k['tflops_algorithmic']:.0f} ({k['frac_mfma']:.2f}) | "
                        f"{k['hbm_gbs_algorithmic']:.0f} ({k['frac_hbm']:.2f}) | {t.get('hbm_bytes_per_step', 0) / 1e9:.1f} ({t.get('algorithmic_bytes_per_step', 0) / 1e9:.1f}) |")
        stats = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
        if os.path.exists(stats):
            avg = {r["kernel"]: float(r["avg_ms"]) for r in csv.DictReader(open(stats))}
            pick = lambda key: next((v for k, v in avg.items() if key in k), None)
            rows.append("")
            fw = (f"`mlp_wide_fwd_kernel<.., 0>` {pick('mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 0>'):.3f} ms, `mlp_wide_fwd_kernel<.., 1>` "
                  f"{pick('mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 1>'):.3f} ms") if live else f"`mlp_wide_fwd_kernel` {pick('mlp_wide_fwd_kernel'):.3f} ms"
            rows.append(f"`rocprofv3 --kernel-trace --stats` of the same run (`profiles/{tag}_bench_kernel_stats.csv`), average launch: "
                        f"`dw_group_kernel<true,true,1>` {pick('dw_group_kernel<true'):.3f} ms, {fw}, "
                        f"`mlp_wide_bwd_kernel` {pick('mlp_wide_bwd_kernel'):.3f} ms.")
        k = line["kernels"]
        big = sum(v["ms_per_step"] for v in k.values())
        ex = f" ({line['step_tflops_executed']:.0f} TFLOP/s executed = {line['step_frac_mfma_executed']:.2f}: a skipped dead point earns nothing there)" if "step_frac_mfma_executed" in line else ""
        rows.append(f"Whole step: {line['step_tflops_algorithmic']:.0f} TFLOP/s algorithmic = {line['step_frac_mfma']:.2f} of the dense bf16 MFMA peak{ex}; "
                    f"the {len(k)} groups sum to {big:.2f} of the step's {line['ms_per_step']:.2f} ms (kernel-group pass and timed region see "
                    f"different live shares).")
        out["kernels"] = "\n".join(rows)
    # ---- SQ counters
    sq = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.md") if tag else None
    if sq and os.path.exists(sq):
        txt = open(sq).read()
        rows = ["| kernel | non-MFMA instructions per MFMA | matrix pipe busy / wave cycles | issuing | issue-stalled | parked in waits | LDS conflict share |", "|---|---|---|---|---|---|---|"]
        for m in re.finditer(r"## (\S+)([^\n]*)\n(?:.*\n)*?Derived: ([^\n]*)", txt):
            d = m.group(3)
            g = lambda pat: (re.search(pat, d) or [None, "?"])[1]
            cols = [g(r"per MFMA ([\d.]+)"), g(r"wave cycles = ([\d.]+)"), g(r"issuing (\d+%)"), g(r"issue-stalled (\d+%)"),
                    g(r"s_barrier (\d+%)"), g(r"active cycles ([\d.]+%)")]
            rows.append("| `" + m.group(1) + "`" + (", no stash (the pass over all the points)" if "no stash" in m.group(2) else ", stash on" if "stash on" in m.group(2) else "") + " | " + " | ".join(cols) + " |")
        rows.append(f"\n(`profiles/{tag}_sq_counters.md`, `profiles/summarize_sq.py`; per-launch counter values, fine-pass shape.)")
        out["sq"] = "\n".join(rows)
    # ---- launches of one timed step
    st = os.path.join(ROOT, "profiles", f"{tag}_step_trace.txt") if tag else None
    if st and os.path.exists(st):
        head = open(st).read().split("\n")[:2]
        out["step"] = "`profiles/%s_step_trace.txt` (one step of the timed region, `tools/trace_step.py`): %s; %s." % (tag, head[0].strip(), head[1].strip())
    return out


def rewrite(path, blk, check):
    s = open(path).read()
    changed = False
    for name, body in blk.items():
        pat = re.compile(r"(<!-- numbers:%s -->\n)(.*?)(<!-- /numbers:%s -->)" % (name, name), re.S)
        m = pat.search(s)
        if not m:
            continue
        body = body + "\n"
        if m.group(2) != body:
            changed = True
            s = s[:m.start(2)] + body + s[m.end(2):]
    if changed and not check:
        open(path, "w").write(s)
    return changed


if __name__ == "__main__":
    check = "--check" in sys.argv
    blk = blocks()
    stale = [f for f in ("DESIGN.md", "README.md") if rewrite(os.path.join(ROOT, f), blk, check)]
    if check and stale:
        print("out of date (run tools/quoted_numbers.py):", stale)
        sys.exit(1)
    print("up to date" if not stale else f"rewrote {stale}")
