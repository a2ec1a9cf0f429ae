#!/usr/bin/env python3
"""Developer tool: throughput of the eval path: one full 640x1120 pose through NeRFAll.render_path
(forward only, N_samples 64 + N_importance 64, inference=True).  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lush_nerf_amd import synth, ops
import test_gpu_parity as T

dev = torch.device("cuda:0")
H, W, F = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF
K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
for planes in ((2, 2), (ops.PLANES_F16, 1)):
    net = T._model(precision=planes).eval()
    poses = torch.from_numpy(synth.poses(2, 1)).to(dev)
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, near=0., far=1.)
    chunk = int(os.environ.get("CHUNK", 1024 * 64))
    with torch.no_grad():
        net(H, W, K, chunk=chunk, poses=poses[:1], render_kwargs=rk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rgbs, noise, depths = net(H, W, K, chunk=chunk, poses=poses, render_kwargs=rk)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / poses.shape[0]
    print(json.dumps({"planes_fwd": planes[0], "pose": f"{H}x{W}", "chunk": chunk, "s_per_pose": round(dt, 4),
                      "rays_per_s": round(H * W / dt, 1), "finite": bool(torch.isfinite(rgbs).all())}), flush=True)
