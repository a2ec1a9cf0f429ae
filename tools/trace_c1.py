#!/usr/bin/env python3
"""Developer tool: the launches of one replayed BASELINE-config-1 step (bench.py --extra C1) from a rocprofv3 kernel trace.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_c1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --also "" --extra C1
  python tools/trace_c1.py gpurun_out/trace_c1"""
import csv, glob, os, sys
src = sys.argv[1]
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# C1 steps are cut at gen_rays_kernel too; take the last complete one (a graph replay)
cuts = [i for i, r in enumerate(rows) if "gen_rays_kernel" in r["Kernel_Name"]]
a, b = cuts[-2], cuts[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"{len(seg)} launches, {(t1 - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us in kernels")
prev = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("lush::", "").split("(")[0][:64]
    print(f"  +{(s - prev) / 1e3:6.1f} gap {(e - s) / 1e3:7.1f} us  {n}  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}")
    prev = e
