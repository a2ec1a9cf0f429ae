#!/usr/bin/env python3
"""Developer tool: one optimisation step as the GPU saw it, from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --also "" --extra ""
  python tools/trace_step.py gpurun_out/trace [--step -2] [--list]

Steps are cut at `gen_rays_kernel` (the first launch of Trainer.step with device-side ray generation).  Prints, for one step,
the launches in order (--list), the count and time per kernel, the launches outside the six big MLP launches and the time
the GPU spent between kernels."""
import csv
import glob
import os
import sys
from collections import OrderedDict

src = sys.argv[1]
step = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit(f"no *kernel_trace.csv under {src}")
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "gen_rays_kernel" in r["Kernel_Name"]]
if len(cuts) < 3:
    sys.exit("fewer than three steps in the trace")
a, b = cuts[step], (cuts[step + 1] if step + 1 < 0 or step + 1 < len(cuts) else len(rows))
if step == -1:
    b = len(rows)
seg = rows[a:b]
BIG = ("mlp_wide_fwd_kernel", "mlp_wide_bwd_kernel", "dw_group_kernel<true", "mlp_chain_fwd_kernel<lush::NetT<256", "mlp_chain_bwd_kernel<lush::NetT<256",
       "mlp_chain_fwd_half", "mlp_chain_bwd_half")
short = lambda n: n.replace("void ", "").replace("lush::", "").split("(")[0][:70]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
per = OrderedDict()
for r in seg:
    k = short(r["Kernel_Name"])
    d = per.setdefault(k, [0, 0])
    d[0] += 1
    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
big = [r for r in seg if any(x in r["Kernel_Name"] for x in BIG)]
small = [r for r in seg if r not in big]
t_big = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big)
t_small = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small)
print(f"step {step}: {len(seg)} launches, {(t1 - t0) / 1e6:.3f} ms wall, {busy / 1e6:.3f} ms in kernels, {(t1 - t0 - busy) / 1e6:.3f} ms between kernels")
print(f"  big MLP launches: {len(big)} ({t_big / 1e6:.3f} ms); everything else: {len(small)} launches, {t_small / 1e6:.3f} ms in kernels, "
      f"{(t1 - t0 - t_big) / 1e6:.3f} ms of the step's wall time")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:4d} x {k:72s} {t / 1e3:9.1f} us")
if "--list" in sys.argv:
    prev = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"    +{(s - prev) / 1e3:7.1f} us gap  {(e - s) / 1e3:8.1f} us  {short(r['Kernel_Name'])}")
        prev = e
