#!/usr/bin/env python3
"""Developer tool (needs a GPU): distribution of the long-trajectory metrics (tests/test_gpu_parity.py::
test_long_training_trajectory_follows_the_reference) over repeated runs, per precision mode and kernel variant -- what the
bands of that test were set from."""
import os, sys, argparse, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lush_nerf_amd import model as M, ops, synth, lib
from lush_nerf_amd.trainer import Trainer
from tests import util
g = util.golden("train_trajectory_long")
n, Ns, Ni, seed, steps = (int(x) for x in g["meta"])
dev = torch.device("cuda:0")
def run(planes, variant):
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
    pf, pb = ops.parse_planes(planes)
    net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4), precision=ops.Precision(pf, pb, variant))
    w0 = synth.all_weights(30, seed, sharp=True, rbk_scale=2.0e4)
    M.load_reference_weights(net, w0)
    net = net.to(dev)
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, kernel_start_iter=0, allkernel_start_iter=0)
    targets = torch.from_numpy(g["targets"]).to(dev)
    losses = []
    for s in range(steps):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, seed, 30, step=s).items()}
        b["target"] = targets[s]
        d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, seed, step=s).items()}
        losses.append(tr.step(b, s, draws=d))
    losses = np.asarray([float(x) for x in losses]); ref = np.asarray(g["losses"], dtype=np.float64); win = 25
    mg = losses[:steps // win * win].reshape(-1, win).mean(1); mr = ref[:steps // win * win].reshape(-1, win).mean(1)
    dev_w = np.abs(mg - mr) / mr
    wf = dict(net.state_dict())["mlp_fine.rgb_linear.weight"].detach().cpu().double().numpy()
    wr = np.asarray(g["final_rgb_w"], dtype=np.float64); w_init = np.asarray(w0["mlp_fine.rgb_linear.weight"], dtype=np.float64)
    du, dr = (wf - w_init).ravel(), (wr - w_init).ravel()
    cos = float(du @ dr / (np.linalg.norm(du) * np.linalg.norm(dr)))
    fall = losses[-win:].mean() / losses[:win].mean() / (ref[-win:].mean() / ref[:win].mean())
    return cos, dev_w.max(), fall

import statistics as st
for planes, variant, reps in (("h,h", 0, 24), ("h,h", lib.VARIANT_BWD_HALF, 24), ("h,h", lib.VARIANT_FWD_HALF, 24), ("h,h", lib.VARIANT_BWD_HALF | lib.VARIANT_FWD_HALF, 24), ("2,2", 0, 12), ("2,h", 0, 12)):
    res = [run(planes, variant) for _ in range(reps)]
    c = [r[0] for r in res]; d = [r[1] for r in res]; f = [r[2] for r in res]
    print(planes, "variant", variant, f"cos mean {st.mean(c):.4f} sd {st.pstdev(c):.4f} min {min(c):.3f} | dev_w mean {st.mean(d):.3f} max {max(d):.2f} | fall min {min(f):.2f} max {max(f):.2f}", flush=True)
