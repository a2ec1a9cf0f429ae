#!/usr/bin/env python3
"""Developer tool (needs a GPU): per-step GPU time (HIP events, no synchronisation inside the loop) and share of live points of
bench.py's headline configuration over STEPS steps of training on the synthetic targets (profiles/r05_live_points.md).
VARIANT=256 (LUSH_VARIANT_DENSE_BWD): the same steps with the backward over all the points.  POLICY=auto|live|dense."""
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
dense_log = []
for i in range(N):
    net.hooks.live_acc = accs[i]
    tr.step(batches[(5 + i) % 4], 5 + i)
    dense_log.append(bool(tr._dense_now))
    ev[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
sh = [(a[0] + a[2]).item() / max((a[1] + a[3]).item(), 1) for a in accs]
import numpy as np
print(f"steps the policy ran with the dense backward: {sum(1 for d in dense_log if d)} of {N} (first at step {5 + dense_log.index(True) if True in dense_log else None}; thresholds {tr.LIVE_MAX_SHARE} / {tr.LIVE_MIN_SHARE}, probe every {tr.LIVE_PROBE_EVERY})")
print(f"variant {variant}, policy {tr.live_policy}: wall {wall / N * 1e3:.3f} ms/step over {N} steps; first 20 steps {sum(ms[:20]) / 20:.3f} ms/step "
      f"(live share {sum(sh[:20]) / 20:.3f})")
if any(sh):
    A = np.stack([np.ones(N), np.array(sh)], 1)
    (a, b), *_ = np.linalg.lstsq(A, np.array(ms), rcond=None)
    print(f"least squares over the {N} steps: ms/step = {a:.2f} + {b:.2f} x share (residual rms {float(np.sqrt(np.mean((A @ [a, b] - np.array(ms)) ** 2))):.3f} ms)")
print("| steps | ms/step (mean) | live share (mean) | min .. max share |\n|---|---|---|---|")
B = max(N // 20, 1)
for k in range(0, N, B):
    m, q = ms[k:k + B], sh[k:k + B]
    print(f"| {5 + k} .. {5 + k + len(m) - 1} | {sum(m) / len(m):.3f} | {sum(q) / len(q):.3f} | {min(q):.3f} .. {max(q):.3f} |")
dist.destroy_process_group()
