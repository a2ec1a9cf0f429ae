#!/usr/bin/env python3
"""Developer aid (GPU box): render outputs of the headline mode (h,h) against the two-plane forward (2,2) at the BENCH's size (N_rand 4096, blur
kernel on: 20 480 marched rays) on the initialisation field and on the trained-like field -- how the maximum deviation over the batch
and the number of rays beyond 1e-4 behave behind a sharp density head (both runs use the same explicit draws)."""
import sys, os, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lush_nerf_amd import lib, model as M, ops, synth
lib.load()
dev = torch.device("cuda:0")
H, W, F, n_img, n, Ns, Ni = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 30, 4096, 64, 64
args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=Ni, netdepth=8, netwidth=256,
                          netdepth_fine=8, netwidth_fine=256, rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 1000, n_img).items()}
d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, 0).items()}
K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
for label, kw in (("initialisation", {}), ("trained_like", dict(trained_like=True)), ("trained_like x2", dict(trained_like=(6000.0, 40.0)))):
    w = synth.all_weights(n_img, 0, rbk_scale=2.0e4, **kw)
    outs = {}
    for name, prec in (("h,h", ops.Precision(ops.PLANES_F16, ops.PLANES_F16)), ("2,2", ops.Precision(2, 2))):
        net = M.NeRFAll(args, M.RBK(n_img, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4), precision=prec)
        M.load_reference_weights(net, w)
        net = net.to(dev).train()
        with torch.no_grad():
            out = net(H, W, K, chunk=1 << 20, rays=b["rays"], rays_info={"images_idx": b["images_idx"]}, retraw=True, force_naive=False, allkernel=True,
                      kernel_pixel=b["fq_mask"], perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False,
                      near=0., far=1., draws=d)
        outs[name] = [out[i].float().cpu() for i in (0, 5)]
    for oname, i in (("rgb_blur", 0), ("rgb (sharp)", 1)):
        a, r = outs["h,h"][i], outs["2,2"][i]
        e = ((a - r).abs().amax(-1) / r.abs().max())
        print(f"{label}: {oname}: {e.numel()} rays, max {float(e.max()):.2e}, 99.9th percentile {float(e.kthvalue(int(e.numel() * 0.999))[0]):.2e}, median {float(e.median()):.2e}, "
              f"rays beyond 1e-4: {int((e > 1e-4).sum())}", flush=True)
