#!/usr/bin/env python3
"""Developer tool (needs a GPU): how long the HOST takes to enqueue one Trainer.step of the headline configuration (no
synchronisation inside the loop) against the GPU time of the same steps, eagerly and as the replayed HIP graph (step_graph)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench
from lush_nerf_amd import lib, ops, synth
from lush_nerf_amd.trainer import Trainer
lib.load()
for mode in ("eager", "graph"):
    net = bench.make_model(bench.model_args(64), dev, ops.Precision(ops.PLANES_F16, ops.PLANES_F16, 0))
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0, allkernel_start_iter=1 << 30, distributed=True)
    poses = torch.from_numpy(synth.poses(30, 1000)).to(dev)
    batches = []
    for s in range(4):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.pixel_batch(4096, seed=1000, step=s).items()}
        b["c2w"] = poses
        batches.append(b)
    step = tr.step if mode == "eager" else tr.step_graph
    for i in range(6):
        step(batches[i % 4], i)
    torch.cuda.synchronize()
    N = 40
    t0 = time.perf_counter()
    host = []
    for i in range(N):
        h0 = time.perf_counter()
        step(batches[i % 4], 6 + i)
        host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    host.sort()
    print(f"{mode}: host enqueue {t_enq / N * 1e3:.2f} ms/step (median {host[N // 2] * 1e3:.2f}), wall incl. GPU {t_all / N * 1e3:.2f} ms/step", flush=True)
    del tr, net
    torch.cuda.empty_cache()
dist.destroy_process_group()
