#!/usr/bin/env python3
"""Developer aid (needs a GPU): op-level torch.profiler view of ONE training step of BASELINE config 2 -- which torch
ops (copies, fills, elementwise) still surround the HIP kernels, with their Python call sites.  Not a test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lush_nerf_amd import ops, synth
from lush_nerf_amd.trainer import Trainer

dev = torch.device("cuda:0")
n_rand = int(os.environ.get("N_RAND", 4096))
net = bench.make_model(bench.model_args(64), dev, ops.Precision(*ops.parse_planes(os.environ.get("LUSH_PLANES", "h,h"))))
tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0, allkernel_start_iter=1 << 30)
poses = torch.from_numpy(synth.poses(30, 1000)).to(dev)
bs = []
for s in range(3):
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.pixel_batch(n_rand, seed=1000, step=s).items()}
    b["c2w"] = poses
    bs.append(b)
for i in range(2):
    tr.step(bs[i], i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(bs[2], 2)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=60, max_name_column_width=50, max_src_column_width=90))
