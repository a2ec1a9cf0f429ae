#!/usr/bin/env python3
"""Developer tool (needs a GPU): which torch-level ops launch the small copy / fill kernels of one training step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from lush_nerf_amd import ops, synth
from lush_nerf_amd.trainer import Trainer
dev = torch.device("cuda:0")
net = bench.make_model(bench.model_args(64), dev, ops.Precision(*ops.parse_planes("h,h")))
tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0, allkernel_start_iter=1 << 30)
n = int(os.environ.get("N_RAND", 4096))
pb = synth.pixel_batch(n, 1, 30)
b = {k: torch.from_numpy(v).to(dev) for k, v in pb.items()}
b["c2w"] = torch.from_numpy(synth.poses(30, 1)).to(dev)
for i in range(3):
    tr.step(b, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(b, 3)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 or "aten::" in e.key]
for k, c, t in sorted(rows, key=lambda r: -r[1])[:45]:
    print(f"{c:4d} x {k[:90]:90s} device {t:.0f} us")
