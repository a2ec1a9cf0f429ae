#!/usr/bin/env python3
"""Developer tool (needs a GPU): per-tensor gradient errors of every precision mode, where the bench runs them.

  python tools/parity_table.py > gpurun_out/r05_parity.jsonl          (on the GPU box; ~6 min)
  python tools/parity_table.py --summarize gpurun_out/r05_parity.jsonl > profiles/r05_parity_summary.md

For each mode in MODES: the masked float64 / fp32 oracle pair of tests/gpu_diag.py (masked_grad_check) on the reference fixtures
(t_train_e2e, t_consistency, t_lindisp_white, t_train_c1, t_consist_step) and at the bench's own regime (t_train_bench_regime at
64 + 64 and 128 + 128), one JSON line per (mode, case, parameter tensor): e_gpu = |g_gpu - g_f64| / |g_f64|, e_f32 = the fp32
oracle's own."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = os.environ.get("MODES", "h,h;2,h;h,1;2,2").split(";")
CASES = os.environ.get("CASES", "")      # substring filter on the case names (re-run one case)


def collect():
    import torch
    from lush_nerf_amd import lib, ops
    lib.load()
    from tests import gpu_diag as D
    for mode in MODES:
        D.E2E_PLANES = ops.parse_planes(mode)
        for name, fn in (("fixtures: training steps", D.t_train_e2e), ("fixture: consistency", D.t_consistency),
                         ("fixture: lindisp + white", D.t_lindisp_white), ("fixture: config-1 step", D.t_train_c1),
                         ("fixture: combined consistency step", D.t_consist_step),
                         ("bench regime 64+64 (N_rand 512)", lambda: D.t_train_bench_regime(512, 21, 64, 64)),
                         ("bench regime 128+128 (N_rand 256)", lambda: D.t_train_bench_regime(256, 21, 128, 128))):
            if CASES and CASES not in name:
                continue
            D.RESULTS.clear()
            D.PER_TENSOR.clear()
            print(f"== {mode} {name}", file=sys.stderr, flush=True)
            fn()
            torch.cuda.synchronize()
            bad = [r[0] for r in D.RESULTS if not r[3]]
            for tag, k, e_gpu, e_f32 in D.PER_TENSOR:
                print(json.dumps({"mode": mode, "case": name, "tag": tag, "tensor": k, "e_gpu": e_gpu, "e_f32": e_f32}), flush=True)
            outs = {r[0]: r[1] for r in D.RESULTS if any(r[0].endswith(s) for s in ("rgb_blur", "rgb0_blur", "rgb_map", "rgb0", "rgb (tone-mapped)", "rgb (sharp)"))}
            print(json.dumps({"mode": mode, "case": name, "checks": len(D.RESULTS), "failed": bad, "output_errors": outs}), flush=True)


def summarize(path):
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    last = {}
    for i, r in enumerate(rows):      # a (mode, case) that was run again replaces its earlier lines
        if "checks" in r:
            last[(r["mode"], r["case"])] = i
    keep, start = [], 0
    for i, r in enumerate(rows):
        if "checks" in r:
            if last[(r["mode"], r["case"])] == i:
                keep += rows[start:i + 1]
            start = i + 1
    rows = keep
    per = [r for r in rows if "tensor" in r]
    meta = [r for r in rows if "checks" in r]
    modes = []
    for r in meta:
        if r["mode"] not in modes:
            modes.append(r["mode"])
    cases = []
    for r in meta:
        if r["case"] not in cases:
            cases.append(r["case"])
    out = ["# Round 5: gradient error per parameter tensor and precision mode (MI355X)\n",
           "Written by `tools/parity_table.py --summarize` from `tools/parity_table.py`'s run on a GPU box (raw lines: gpurun_out/, scratch).",
           "e_gpu = max|g_gpu - g_f64| / max|g_f64| per parameter tensor, g_f64 = the oracle in float64 evaluated with the ReLU decisions",
           "the GPU took (tests/gpu_diag.py `masked_grad_check`); e_f32 = the same for the fp32 oracle, i.e. the reference's own arithmetic.",
           "A tensor is *well-conditioned* when e_f32 < 1e-5 (fp32 itself reproduces float64); the others are cancelling sums",
           "(1-element biases, the alpha head), where every mode's error is the condition number times its operand rounding.",
           "The gradients are the PRODUCT's: in the training-step fixtures and the bench regime those of the live-point backward (the run that",
           "keeps every point's stash supplies the ReLU decisions only); in the other cases the stash-keeping run's own (the dense form, which",
           "the live-point backward reproduces to 1e-7 L2: tests/test_gpu_parity.py::test_live_point_march_*).  This table: regenerated after the",
           "live-point backward went in (raw lines: profiles/r05_parity_raw.jsonl.gz); every check of every case passed in all four modes.\n",
           "## Worst tensor per case: well-conditioned tensors / all tensors\n",
           "| case | " + " | ".join(f"({m})" for m in modes) + " |", "|---|" + "---|" * len(modes)]
    for c in cases:
        cells = []
        for m in modes:
            t = [r for r in per if r["mode"] == m and r["case"] == c]
            well = [r["e_gpu"] for r in t if r["e_f32"] < 1e-5]
            cells.append(f"{max(well):.1e} / {max(r['e_gpu'] for r in t):.1e}" if t else "-")
        out.append(f"| {c} | " + " | ".join(cells) + " |")
    out += ["", "## Render-output error of the same runs (worst of the colour outputs; north-star bound 1e-4)\n",
            "| case | " + " | ".join(f"({m})" for m in modes) + " |", "|---|" + "---|" * len(modes)]
    for c in cases:
        cells = []
        for m in modes:
            mm = [r for r in meta if r["mode"] == m and r["case"] == c]
            cells.append(f"{max(mm[0]['output_errors'].values()):.1e}" if mm and mm[0]["output_errors"] else "-")
        out.append(f"| {c} | " + " | ".join(cells) + " |")
    # tensors above 1e-2 in the headline mode: which operand rounding is it?
    head = modes[0]
    big = sorted({(r["case"], r["tensor"]) for r in per if r["mode"] == head and r["e_gpu"] > 1e-2})
    out += ["", f"## Tensors above 1e-2 in ({head}): the same tensor in the other modes\n",
            "If the forward's operand rounding is what the condition number amplifies, the modes that share the forward share the error,",
            "whatever their backward.\n",
            "| case | tensor | fp32 oracle e_f32 | " + " | ".join(f"({m})" for m in modes) + " |", "|---|---|---|" + "---|" * len(modes)]
    for c, k in big:
        ef = [r["e_f32"] for r in per if r["case"] == c and r["tensor"] == k]
        cells = []
        for m in modes:
            t = [r["e_gpu"] for r in per if r["mode"] == m and r["case"] == c and r["tensor"] == k]
            cells.append(f"{max(t):.1e}" if t else "-")
        out.append(f"| {c} | {k} | {max(ef):.1e} | " + " | ".join(cells) + " |")
    worst = {m: max([r["e_gpu"] for r in per if r["mode"] == m] + [0.0]) for m in modes}
    wellw = {m: max([r["e_gpu"] for r in per if r["mode"] == m and r["e_f32"] < 1e-5] + [0.0]) for m in modes}
    out += ["", "## Worst over every case\n", "| mode | well-conditioned tensors | all tensors | failed checks |", "|---|---|---|---|"]
    for m in modes:
        nf = sum(len(r["failed"]) for r in meta if r["mode"] == m)
        out.append(f"| ({m}) | {wellw[m]:.1e} | {worst[m]:.1e} | {nf} |")
    print("\n".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        collect()
