#!/usr/bin/env python3
"""Developer tool (needs a GPU): per-step GPU time (HIP events, no synchronisation inside the loop) and live share of bench.py's
headline configuration, to see what a fresh box does to the first steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench
from lush_nerf_amd import lib, ops, synth
from lush_nerf_amd.trainer import Trainer
lib.load()
variant = int(os.environ.get("VARIANT", 0))
net = bench.make_model(bench.model_args(64), dev, ops.Precision(ops.PLANES_F16, ops.PLANES_F16, variant))
tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0, allkernel_start_iter=1 << 30, distributed=True)
poses = torch.from_numpy(synth.poses(30, 1000)).to(dev)
batches = []
for s in range(4):
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.pixel_batch(4096, seed=1000, step=s).items()}
    b["c2w"] = poses
    batches.append(b)
N = int(os.environ.get("STEPS", 40))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
accs = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(N)]
tr.live_policy = os.environ.get("POLICY", "auto")
for i in range(5):
    tr.step(batches[i % 4], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    net.hooks.live_acc = accs[i]
    tr.step(batches[(5 + i) % 4], 5 + i)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
sh = [(a[0] + a[2]).item() / max((a[1] + a[3]).item(), 1) for a in accs]
print(f"variant {variant}: wall {wall / N * 1e3:.2f} ms/step; first 20 steps {sum(ms[:20]) / 20:.2f} ms")
print(" ".join(f"{m:.1f}({s:.2f})" for m, s in zip(ms, sh)))
dist.destroy_process_group()
