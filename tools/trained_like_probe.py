#!/usr/bin/env python3
"""Developer aid (GPU box): live-point share and step time of BASELINE config 2 from `synth.all_weights(trained_like=(scale, bias))`
density fields -- how bench.py's `trained_like` workload was calibrated (coarse pass ~0.1 live, the fine pass what sample_pdf makes of it).
  python tools/trained_like_probe.py "1500,-10;1500,-6;3000,-20"
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench
from lush_nerf_amd import model as M, ops, synth
from lush_nerf_amd.trainer import Trainer

dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pairs = [tuple(float(x) for x in p.split(",")) for p in (sys.argv[1] if len(sys.argv) > 1 else "1500,-10").split(";")]
poses = torch.from_numpy(synth.poses(30, 1000)).to(dev)
batches = []
for s in range(4):
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.pixel_batch(4096, seed=1000, step=s).items()}
    b["c2w"] = poses
    batches.append(b)
for tl in [None] + pairs:
    rbk = M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4)
    net = M.NeRFAll(bench.model_args(64), rbk, precision=ops.Precision(ops.PLANES_F16, ops.PLANES_F16))
    M.load_reference_weights(net, synth.all_weights(30, 0, trained_like=tl if tl else False))
    net = net.to(dev)
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0, allkernel_start_iter=1 << 30, distributed=True)
    tr.live_policy = "live"
    tr.lrate = float(os.environ.get("LRATE", 0.0)) if tl else tr.lrate      # the field is HELD: every kernel of the step runs, Adam with a zero rate
    for i in range(3):
        tr.step(batches[i % 4], i)
    torch.cuda.synchronize()
    c0 = tr.live_counts()
    t0 = time.perf_counter()
    for i in range(10):
        tr.step(batches[(3 + i) % 4], 3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    c1 = tr.live_counts()
    fl, fa, cl, ca = (b - a for a, b in zip(c0, c1))
    print(json.dumps({"trained_like": tl, "ms_per_step": round(dt * 1e3, 3), "share": round((fl + cl) / (fa + ca), 4), "fine": round(fl / fa, 4),
                      "coarse": round(cl / ca, 4), "faults": tr.faults()}), flush=True)
    del tr, net
    torch.cuda.empty_cache()
dist.destroy_process_group()
