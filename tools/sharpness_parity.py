import sys, argparse, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lush_nerf_amd import lib, model as M, ops, synth
from oracle import lush_oracle as O
lib.load()
dev = torch.device("cuda:0")
H, W, F, n_img, n, Ns, Ni, seed = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 30, 64, 64, 64, 0
args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=Ni, netdepth=8, netwidth=256,
                          netdepth_fine=8, netwidth_fine=256, rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
for label, kw in (("init", {}), ("sharp x400", dict(sharp=True)), ("trained_like", dict(trained_like=True)), ("trained_like x2", dict(trained_like=(6000.0, 40.0)))):
    w = synth.all_weights(n_img, seed, rbk_scale=2.0e4, **kw)
    b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(n, 1000, n_img).items()}
    d = {k: torch.from_numpy(v) for k, v in synth.draws(n * 5, Ns, Ni, seed).items()}
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    p = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    with torch.no_grad():
        ref = O.forward_train(p, H, W, F, b["rays"], b["images_idx"], Ns, Ni, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws=d)
        p64 = {k: v.double() for k, v in p.items()}
        ref64 = O.forward_train(p64, H, W, F, b["rays"].double(), b["images_idx"], Ns, Ni, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws={k: v.double() for k, v in d.items()})
    line = [label, "fp32 oracle vs float64: %.1e" % float((ref[5] - ref64[5].float()).abs().max() / ref64[5].abs().max())]
    for name, prec in (("h,h", ops.Precision(ops.PLANES_F16, ops.PLANES_F16)), ("2,2", ops.Precision(2, 2))):
        net = M.NeRFAll(args, M.RBK(n_img, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4), precision=prec)
        M.load_reference_weights(net, w)
        net = net.to(dev).train()
        with torch.no_grad():
            out = net(H, W, K, chunk=1 << 20, rays=b["rays"].to(dev), rays_info={"images_idx": b["images_idx"].to(dev)}, retraw=True, force_naive=False, allkernel=True,
                      kernel_pixel=b["fq_mask"].to(dev), perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False,
                      near=0., far=1., draws={k: v.to(dev) for k, v in d.items()})
        errs = {nm: float((out[i].cpu() - ref[i]).abs().max() / ref[i].abs().max()) for nm, i in (("rgb_blur", 0), ("rgb0_blur", 1), ("rgb", 5), ("rgb0", 6))}
        line.append(name + " " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    print(" | ".join(line), flush=True)
