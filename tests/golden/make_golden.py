#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference at /root/reference is imported read-only (sys.dont_write_bytecode
is forced; cv2/imageio, which the hot path never touches, are registered as
empty modules exactly as SURVEY.md section 8c records).  Weights, rays and random
draws come from the build-owned generator lush_nerf_amd/synth.py; the reference's
torch.rand / torch.randn_like calls are served from those draws in call order
(shape-checked), so fixtures are independent of torch's RNG streams.

Only DATA is written: inputs are regenerated from (seed, shape) recorded in the
fixture; outputs are the reference's tensors.  Gradients of tensors > 4096
elements are stored as 8 fixed projections + the L2 norm.

Render_Aligned_Pixel (the consistency branch) calls ``.cuda()`` on three tensors it creates and
draws its anchor / pixel samples from ``random`` / ``np.random``; for that one case
``torch.Tensor.cuda`` is an identity stand-in (there is no GPU in the build container) and the two
host draws are served from values recorded in the fixture.  The arithmetic is the reference's.
"""
import os
import sys
import types
import argparse

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch

from lush_nerf_amd import synth

for _m in ("cv2", "imageio"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.path.insert(0, REF)
import models.lushnerf as ref_model          # noqa: E402
import utils.run_lushnerf_helpers as ref_helpers  # noqa: E402

H, W, FOCAL = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF
K = [[FOCAL, 0, W / 2], [0, FOCAL, H / 2], [0, 0, 1]]
NUM_IMG = 30


def build_ref(N_importance, weights, rmnear=80):
    args = argparse.Namespace(
        blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
        N_importance=N_importance, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
        rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
        render_rmnearplane=rmnear)
    rbk = ref_model.RBK(num_img=NUM_IMG, view_embed_ch=64, D_rbk=4, W_rbk=64, skips_rbk=[4],
                        num_motion_rbk=4, D_rbk_r=1, W_rbk_r=32, output_ch_rbk_r=3,
                        D_rbk_v=1, W_rbk_v=32, output_ch_rbk_v=3, D_rbk_w=1, W_rbk_w=32,
                        rbk_se_rv_window=0.1, use_dpnerf=True, rbk_use_origin=True)
    net = ref_model.NeRFAll(args, rbk)
    sd = net.state_dict()
    new = {}
    for k in sd:
        canon = k
        if k.startswith("blur_kernel_net.RBK."):
            canon = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        elif k.startswith("blur_kernel_net.view_embed_layer.") or k.startswith("dbk_view_embedding."):
            canon = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        if canon in weights:
            new[k] = torch.from_numpy(weights[canon].copy())
        elif N_importance == 0 and canon.startswith("mlp_fine."):
            continue
        else:
            raise KeyError(k)
    net.load_state_dict(new, strict=True)
    return net


class ServeDraws:
    """Serve torch.rand / torch.randn_like from a list of numpy arrays."""

    def __init__(self, seq):
        self.seq = [torch.from_numpy(np.ascontiguousarray(a)) for a in seq]
        self.i = 0

    def _next(self, shape):
        t = self.seq[self.i]
        assert tuple(t.shape) == tuple(shape), (self.i, t.shape, shape)
        self.i += 1
        return t.clone()

    def __enter__(self):
        self._rand, self._randn_like = torch.rand, torch.randn_like
        torch.rand = lambda *s, **k: self._next(s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
        torch.randn_like = lambda x, **k: self._next(x.shape)
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn_like = self._rand, self._randn_like
        assert self.i == len(self.seq), (self.i, len(self.seq))


def proj_vecs(name, numel):
    return synth.normal((8, numel), 1234, synth._stream("proj." + name)).astype(np.float64)


def pack_grads(named_grads):
    out = {}
    for k, g in named_grads.items():
        g = g.detach().numpy()
        if g.size <= 4096:
            out["grad." + k] = g
        else:
            out["gradproj." + k] = (proj_vecs(k, g.size) @ g.reshape(-1).astype(np.float64))
            out["gradnorm." + k] = np.array(np.linalg.norm(g.astype(np.float64)))
    return out


def canon_name(k):
    if k.startswith("blur_kernel_net.RBK."):
        return "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
    if k.startswith("blur_kernel_net.view_embed_layer."):
        return "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
    return k


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def case_train(name, n_rand, Ns, Ni, force_naive, sharp, seed, allkernel=True):
    wts = synth.all_weights(NUM_IMG, seed, sharp=sharp, rbk_scale=2.0e4 if not force_naive else 1.0)
    net = build_ref(Ni, wts)
    net.train()
    b = synth.ray_batch(n_rand, seed, NUM_IMG)
    M = 1 if force_naive else 5
    d = synth.draws(n_rand * M, Ns, Ni, seed)
    seq = [d["t_rand"], d["noise_c"]] + ([d["u"], d["noise_f"]] if Ni > 0 else [])
    rays = torch.from_numpy(b["rays"]).requires_grad_(True)
    info = {"images_idx": torch.from_numpy(b["images_idx"])}
    kw = dict(perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=1., inference=False, near=0., far=1.)
    with ServeDraws(seq):
        out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info=info, retraw=True,
                  force_naive=force_naive, allkernel=allkernel,
                  kernel_pixel=torch.from_numpy(b["fq_mask"]).bool(), **kw)
    rgb_blur, rgb0_blur, _, noise, _, rgb, rgb0 = out
    target = torch.from_numpy(b["target"])
    loss = ref_helpers.img2mse(rgb_blur, target) * 0.5 + ref_helpers.img2l1(rgb_blur, target) * 0.5 \
        + ref_helpers.img2mse(rgb0_blur, target) * 0.5 + ref_helpers.img2l1(rgb0_blur, target) * 0.5
    loss.backward()
    grads = {}
    seen = set()
    for k, p in net.named_parameters():
        ck = canon_name(k)
        if ck in seen:
            continue
        seen.add(ck)
        if p.grad is not None:
            grads[ck] = p.grad
    arrs = dict(meta=np.array([n_rand, Ns, Ni, int(force_naive), int(sharp), seed, int(allkernel)]),
                rgb_blur=rgb_blur.detach().numpy(), rgb0_blur=rgb0_blur.detach().numpy(),
                noise=noise.detach().numpy(), loss=np.array(loss.item()),
                grad_rays=rays.grad.numpy() if rays.grad is not None else np.zeros(0))
    if not force_naive:
        arrs.update(rgb=rgb.detach().numpy(), rgb0=rgb0.detach().numpy())
    arrs.update(pack_grads(grads))
    arrs["grad_none"] = np.array(sorted(set(canon_name(k) for k, p in net.named_parameters()
                                            if p.grad is None)))
    save(name, **arrs)


def case_render_rays(name, n_rays, Ns, Ni, train, sharp, seed):
    """render_infer -> render_rays: the per-ray dict (rgb_map, depth, acc, density, raw, ...)."""
    wts = synth.all_weights(NUM_IMG, seed, sharp=sharp)
    net = build_ref(Ni, wts)
    net.train(train)
    b = synth.ray_batch(n_rays, seed, NUM_IMG)
    d = synth.draws(n_rays, Ns, Ni, seed)
    seq = ([d["t_rand"], d["noise_c"]] + ([d["u"], d["noise_f"]] if Ni > 0 else [])) if train else []
    kw = dict(perturb=1. if train else 0., N_importance=Ni, N_samples=Ns, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=1. if train else 0., inference=not train,
              near=0., far=1., retraw=True)
    with torch.no_grad(), ServeDraws(seq):
        (rgb, depth, acc, extras), noise = net.render_infer(
            H, W, K, 1 << 20, rays=torch.from_numpy(b["rays"]), **kw)
    arrs = dict(meta=np.array([n_rays, Ns, Ni, int(train), int(sharp), seed]),
                rgb_map=rgb.numpy(), depth_map=depth.numpy(), acc_map=acc.numpy(),
                noise_rgb=noise.numpy())
    for k, v in extras.items():
        arrs[k] = v.numpy()
    save(name, **arrs)


def case_sample_pdf(seed=5):
    R, S, Ni = 48, 64, 64
    bins = np.sort(synth.uniform((R, S - 1), 0, 1, seed, 1), -1)
    w = synth.uniform((R, S - 2), 0, 1, seed, 2) ** 8        # spiky
    w[:4] = 0.0                                              # all-zero weights rows
    w[4:8, 10:] = 0.0
    u = np.minimum(synth.uniform((R, Ni), 0, 1, seed, 3), np.float32(1 - 2 ** -24))
    with ServeDraws([u]):
        s_rand = ref_helpers.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), Ni, det=False)
    s_det = ref_helpers.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), Ni, det=True)
    save("sample_pdf", bins=bins, weights=w, u=u, s_rand=s_rand.numpy(), s_det=s_det.numpy())


def case_rbk(seed=6):
    n = 40
    wts = synth.all_weights(NUM_IMG, seed, rbk_scale=3.0e5)   # rotations up to ~0.5 rad
    net = build_ref(64, wts)
    b = synth.ray_batch(n, seed, NUM_IMG)
    with torch.no_grad():
        new_rays, ccw = net.mlp_rbk(torch.from_numpy(b["rays"]),
                                    {"images_idx": torch.from_numpy(b["images_idx"])})
        o, d = ref_helpers.ndc_rays(H, W, FOCAL, 1., new_rays[..., 0], new_rays[..., 1])
    save("rbk", meta=np.array([n, seed]), new_rays=new_rays.numpy(), ccw=ccw.numpy(),
         ndc_o=o.numpy(), ndc_d=d.numpy())


def case_checkpoint_layout():
    """Key names / shapes of the reference checkpoint (run_lushnerf.py:687-694) and the Adam parameter-group
    sizes (:359-371).  Structure only: no weights are stored."""
    net = build_ref(64, synth.all_weights(NUM_IMG, 1))
    dp = torch.nn.DataParallel(net, [])
    sd = dp.state_dict()
    noise = list(dp.module.mlp_noise_coarse.parameters())
    ids = set(map(id, noise))
    base = [p for p in dp.parameters() if id(p) not in ids]
    opt = torch.optim.Adam([{"params": base}, {"params": noise, "lr": 5e-4}], lr=5e-4)
    osd = opt.state_dict()
    save("checkpoint_layout", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         group_sizes=np.array([len(g["params"]) for g in osd["param_groups"]]),
         group0_numel=np.array([p.numel() for p in base]), group1_numel=np.array([p.numel() for p in noise]))


def case_sample_pdf_z(seed=8):
    """sample_pdf on bins that ARE the mid-points of a stored z (what render_rays feeds it,
    models/lushnerf.py:435-437): lets the GPU kernel, which takes z and weights, run the reference's vectors."""
    R, S, Ni = 40, 64, 64
    z = np.sort(synth.uniform((R, S), 0, 1, seed, 1), -1)
    w = synth.uniform((R, S), 0, 1, seed, 2) ** 8
    w[:3] = 0.0
    w[3:6, 12:] = 0.0
    u = np.minimum(synth.uniform((R, Ni), 0, 1, seed, 3), np.float32(1 - 2 ** -24))
    zt = torch.from_numpy(z)
    bins = .5 * (zt[..., 1:] + zt[..., :-1])
    wt = torch.from_numpy(w)[..., 1:-1]
    with ServeDraws([u]):
        s_rand = ref_helpers.sample_pdf(bins, wt, Ni, det=False)
    s_det = ref_helpers.sample_pdf(bins, wt, Ni, det=True)
    save("sample_pdf_z", z=z, weights=w, u=u, s_rand=s_rand.numpy(), s_det=s_det.numpy(),
         merged_rand=torch.sort(torch.cat([zt, s_rand], -1), -1)[0].numpy(),
         merged_det=torch.sort(torch.cat([zt, s_det], -1), -1)[0].numpy())


def nondc_batch(n, seed):
    """[n,11] ray batch for the no_ndc configuration (run_lushnerf.py:397-399): world-space o, d, near 2, far 6."""
    b = synth.ray_batch(n, seed, NUM_IMG)
    o, d = b["rays"][..., 0], b["rays"][..., 1]
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    ones = np.ones((n, 1), np.float32)
    return np.concatenate([o, d, 2.0 * ones, 6.0 * ones, vd], -1).astype(np.float32)


def case_lindisp_white(seed=14):
    """render_rays with lindisp=True and white_bkgd=True (models/lushnerf.py:393-396, 349-350): only reachable
    with --no_ndc (near > 0), so the ray batch is fed to render_rays directly."""
    n, Ns, Ni = 40, 64, 64
    wts = synth.all_weights(NUM_IMG, seed, sharp=True)
    net = build_ref(Ni, wts)
    net.train()
    d = synth.draws(n, Ns, Ni, seed)
    batch = torch.from_numpy(nondc_batch(n, seed))
    with torch.no_grad(), ServeDraws([d["t_rand"], d["noise_c"], d["u"], d["noise_f"]]):
        ret, ret_noise = net.render_rays(batch, N_samples=Ns, retraw=True, lindisp=True, perturb=1.,
                                         N_importance=Ni, white_bkgd=True, raw_noise_std=1.)
    arrs = dict(meta=np.array([n, Ns, Ni, seed]), noise_rgb=ret_noise["rgb_map"].numpy())
    for k, v in ret.items():
        arrs[k] = v.numpy()
    save("rays_lindisp_white", **arrs)


def case_eval_forward(seed=15):
    """NeRFAll.forward(poses=...) in eval mode (models/lushnerf.py:671-677 -> render_path :868-896) on a tiny
    image: tone-mapped rgbs, 0.1*sigmoid noise image, depths."""
    H_, W_, F_ = 12, 20, 17.5
    K_ = [[F_, 0, W_ / 2], [0, F_, H_ / 2], [0, 0, 1]]
    wts = synth.all_weights(NUM_IMG, seed, sharp=True)
    net = build_ref(64, wts)
    net.eval()
    poses = torch.from_numpy(synth.poses(2, seed))
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, near=0., far=1.)
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        rgbs, noise, depths = net(H_, W_, K_, chunk=128, poses=poses, render_kwargs=rk)
    save("eval_forward", meta=np.array([H_, W_, seed]), focal=np.array(F_), rgbs=rgbs.numpy(), noise=noise.numpy(),
         depths=depths.numpy())


def case_consistency(seed=31):
    """The consistency branch (run_lushnerf.py:629-650; models/lushnerf.py:664-668, 949-989):
    forward(consist_loss=True) in train mode + compute_mean_with_confidence + the masked L1, with gradients."""
    import random as pyrandom
    V, ns = 5, 32
    wts = synth.all_weights(NUM_IMG, seed, sharp=True)
    net = build_ref(64, wts)
    net.train()
    poses = torch.from_numpy(synth.poses(V, seed))
    HW = H * W
    anchor = 3
    samples = (synth.uniform01(ns, seed, 11) * HW).astype(np.int64)
    # matched pixels of the anchor view in every view: in-image, plus a few that must be clamped
    ax = synth.uniform((V, ns), -40, W + 40, seed, 12)
    ay = synth.uniform((V, ns), -30, H + 30, seed, 13)
    cert_f = synth.uniform((V, ns), 0, 1, seed, 14)
    cert_f[cert_f < 0.35] = 0.0                       # Align_Mask is torch.bool (run_lushnerf.py:292): nonzero -> True
    cert_f[:, 5] = 0.0                                # one pixel no view is certain about (count 0 -> mean 0)
    Align_Matrix = torch.zeros(V, V, HW, 4)
    Align_Mask = torch.zeros(V, V, HW, dtype=torch.bool)
    st = torch.from_numpy(samples)
    Align_Matrix[anchor][:, st, 2] = torch.from_numpy(ax)
    Align_Matrix[anchor][:, st, 3] = torch.from_numpy(ay)
    Align_Mask[anchor][:, st] = torch.from_numpy(cert_f) != 0   # what copy_ of a float certainty into the bool table stores
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, save_warped_ray_img=False, near=0., far=1.)
    keep = (torch.Tensor.cuda, pyrandom.randint, np.random.randint)
    torch.Tensor.cuda = lambda self, *a, **k: self
    pyrandom.randint = lambda a, b: anchor
    np.random.randint = lambda lo, hi=None, size=None: samples.copy()
    import contextlib, io
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rgb_align, cert = net(H, W, K, 1 << 20, poses=poses, render_kwargs=rk, render_factor=0,
                                  rays_info=torch.arange(V), consist_loss=True, Align_matrix=Align_Matrix,
                                  Align_mask=Align_Mask)
    finally:
        torch.Tensor.cuda, pyrandom.randint, np.random.randint = keep
    mask = cert >= 0.8
    mean = ref_helpers.compute_mean_with_confidence(rgb_align, cert, 0.8)
    loss_rgb = torch.sum(torch.abs(rgb_align - mean.unsqueeze(0)) * mask.unsqueeze(2)) / len(mask[mask == 1])
    loss_rgb.backward()
    grads, seen = {}, set()
    for k, p in net.named_parameters():
        ck = canon_name(k)
        if ck in seen:
            continue
        seen.add(ck)
        if p.grad is not None:
            grads[ck] = p.grad
    arrs = dict(meta=np.array([V, ns, seed, anchor]), samples=samples, ax=ax, ay=ay, cert_in=cert_f,
                rgb_align=rgb_align.detach().numpy(), certainty=cert.numpy(), mean=mean.detach().numpy(),
                loss_rgb=np.array(loss_rgb.item()))
    arrs.update(pack_grads(grads))
    arrs["grad_none"] = np.array(sorted(set(canon_name(k) for k, p in net.named_parameters() if p.grad is None)))
    save("consistency", **arrs)


def case_trajectory(seed=41, n_rand=32, steps=40):
    """A short training run of the reference itself: the model of case_train stepped by the reference's optimizer
    set-up (run_lushnerf.py:359-371: Adam, two parameter groups) and learning-rate rule (:681-685), fresh rays and
    draws every step.  Stores the loss curve; the GPU modes are compared with it inside a stated band."""
    Ns = Ni = 64
    wts = synth.all_weights(NUM_IMG, seed, sharp=True, rbk_scale=2.0e4)
    net = build_ref(Ni, wts)
    net.train()
    noise = list(net.mlp_noise_coarse.parameters())
    ids = set(map(id, noise))
    base = [p for p in net.parameters() if id(p) not in ids]
    lrate, decay = 5e-4, 250
    opt = torch.optim.Adam([{"params": base}, {"params": noise, "lr": lrate}], lr=lrate)
    kw = dict(perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1.,
              inference=False, near=0., far=1.)
    losses, global_step = [], 0
    for s in range(steps):
        b = synth.ray_batch(n_rand, seed, NUM_IMG, step=s)
        d = synth.draws(n_rand * 5, Ns, Ni, seed, step=s)
        with ServeDraws([d["t_rand"], d["noise_c"], d["u"], d["noise_f"]]):
            out = net(H, W, K, chunk=1 << 20, rays=torch.from_numpy(b["rays"]),
                      rays_info={"images_idx": torch.from_numpy(b["images_idx"])}, retraw=True, force_naive=False,
                      allkernel=False, kernel_pixel=torch.from_numpy(b["fq_mask"]).bool(), **kw)
        target = torch.from_numpy(b["target"])
        loss = ref_helpers.img2mse(out[0], target) * 0.5 + ref_helpers.img2l1(out[0], target) * 0.5 \
            + ref_helpers.img2mse(out[1], target) * 0.5 + ref_helpers.img2l1(out[1], target) * 0.5
        opt.zero_grad()
        loss.backward()
        opt.step()
        new_lrate = lrate * (0.1 ** (global_step / (decay * 1000)))
        for g in opt.param_groups:
            g["lr"] = new_lrate
        global_step += 1
        losses.append(loss.item())
        print("traj step", s, loss.item(), flush=True)
    sd = {canon_name(k): v for k, v in net.state_dict().items()}
    save("train_trajectory", meta=np.array([n_rand, Ns, Ni, seed, steps]), losses=np.array(losses),
         final_norms=np.array([float(sd[k].double().norm()) for k in sorted(sd)]),
         final_keys=np.array(sorted(sd)),
         final_fine_l3=sd["mlp_fine.pts_linears.3.weight"].numpy()[:8, :8].copy(),
         final_rgb_w=sd["mlp_fine.rgb_linear.weight"].numpy().copy())



def case_trajectory_long(seed=43, n_rand=32, steps=300):
    """A LONGER training run of the reference itself (run_lushnerf.py:603-685), on a problem whose loss can actually fall:
    the targets are what a second, fixed weight set ("teacher", seed + 100) renders for the same rays -- its blurred,
    tone-mapped colour NeRFAll.forward()[0] without jitter or density noise -- instead of U(0,1) colours.  Same optimizer
    set-up and learning-rate rule as case_trajectory, fresh rays and draws every step.  Stores the targets (so the GPU
    test needs no teacher), the loss curve and the final fine rgb head."""
    Ns = Ni = 64
    net = build_ref(Ni, synth.all_weights(NUM_IMG, seed, sharp=True, rbk_scale=2.0e4))
    teacher = build_ref(Ni, synth.all_weights(NUM_IMG, seed + 100, sharp=True, rbk_scale=2.0e4))
    net.train()
    teacher.train()
    noise = list(net.mlp_noise_coarse.parameters())
    ids = set(map(id, noise))
    base = [p for p in net.parameters() if id(p) not in ids]
    lrate, decay = 5e-4, 250
    opt = torch.optim.Adam([{"params": base}, {"params": noise, "lr": lrate}], lr=lrate)
    kw = dict(perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1.,
              inference=False, near=0., far=1.)
    kw_t = dict(kw, perturb=0., raw_noise_std=0.)
    losses, targets, global_step = [], [], 0
    for s in range(steps):
        b = synth.ray_batch(n_rand, seed, NUM_IMG, step=s)
        d = synth.draws(n_rand * 5, Ns, Ni, seed, step=s)
        rays = torch.from_numpy(b["rays"])
        info = {"images_idx": torch.from_numpy(b["images_idx"])}
        mask = torch.from_numpy(b["fq_mask"]).bool()
        with torch.no_grad():
            target = teacher(H, W, K, chunk=1 << 20, rays=rays, rays_info=info, retraw=True, force_naive=False, allkernel=False,
                             kernel_pixel=mask, **kw_t)[0].clone()
        with ServeDraws([d["t_rand"], d["noise_c"], d["u"], d["noise_f"]]):
            out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info=info, retraw=True, force_naive=False, allkernel=False,
                      kernel_pixel=mask, **kw)
        loss = ref_helpers.img2mse(out[0], target) * 0.5 + ref_helpers.img2l1(out[0], target) * 0.5 \
            + ref_helpers.img2mse(out[1], target) * 0.5 + ref_helpers.img2l1(out[1], target) * 0.5
        opt.zero_grad()
        loss.backward()
        opt.step()
        new_lrate = lrate * (0.1 ** (global_step / (decay * 1000)))
        for g in opt.param_groups:
            g["lr"] = new_lrate
        global_step += 1
        losses.append(loss.item())
        targets.append(target.numpy().copy())
        if s % 10 == 0:
            print("long traj step", s, loss.item(), flush=True)
    sd = {canon_name(k): v for k, v in net.state_dict().items()}
    save("train_trajectory_long", meta=np.array([n_rand, Ns, Ni, seed, steps]), losses=np.array(losses),
         targets=np.stack(targets).astype(np.float32),
         final_norms=np.array([float(sd[k].double().norm()) for k in sorted(sd)]), final_keys=np.array(sorted(sd)),
         final_rgb_w=sd["mlp_fine.rgb_linear.weight"].numpy().copy(),
         final_rgb_b=sd["mlp_fine.rgb_linear.bias"].numpy().copy())


def _consist_tables(seed, V=5, ns=32, anchor=3):
    """Align_Matrix / Align_Mask of the consistency branch from the build-owned generator (as case_consistency)."""
    HW = H * W
    samples = (synth.uniform01(ns, seed, 11) * HW).astype(np.int64)
    ax = synth.uniform((V, ns), -40, W + 40, seed, 12)
    ay = synth.uniform((V, ns), -30, H + 30, seed, 13)
    cert_f = synth.uniform((V, ns), 0, 1, seed, 14)
    cert_f[cert_f < 0.35] = 0.0
    cert_f[:, 5] = 0.0
    Align_Matrix = torch.zeros(V, V, HW, 4)
    Align_Mask = torch.zeros(V, V, HW, dtype=torch.bool)
    st = torch.from_numpy(samples)
    Align_Matrix[anchor][:, st, 2] = torch.from_numpy(ax)
    Align_Matrix[anchor][:, st, 3] = torch.from_numpy(ay)
    Align_Mask[anchor][:, st] = torch.from_numpy(cert_f) != 0
    return samples, ax, ay, cert_f, Align_Matrix, Align_Mask


def _named_grads(net):
    grads, seen = {}, set()
    for k, p in net.named_parameters():
        ck = canon_name(k)
        if ck in seen:
            continue
        seen.add(ck)
        if p.grad is not None:
            grads[ck] = p.grad
    return grads


def case_train_c1(seed=51, n_rand=256, Ns=32):
    """BASELINE config 1 as a TRAINING step (round 5): N_rand 256, 32 + 0 samples, naive.  NeRFAll.forward cannot take
    N_importance = 0 (it indexes extras['rgb0'], models/lushnerf.py:660), so the entry is render_infer (:679-763 -> render_rays
    :354-479) as in the reference's own CPU-runnable case; the loss is run_lushnerf.py:652-661 with rgb0 = rgb (no fine pass: both
    terms see the one colour), backward through the coarse network alone."""
    wts = {k: v for k, v in synth.all_weights(NUM_IMG, seed, sharp=True).items()}
    net = build_ref(0, wts)
    net.train()
    b = synth.ray_batch(n_rand, seed, NUM_IMG)
    d = synth.draws(n_rand, Ns, 0, seed)
    kw = dict(perturb=1., N_importance=0, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False,
              near=0., far=1., retraw=True)
    with ServeDraws([d["t_rand"], d["noise_c"]]):
        (rgb, depth, acc, extras), noise = net.render_infer(H, W, K, 1 << 15, rays=torch.from_numpy(b["rays"]), **kw)
    tm = net.tonemapping(rgb)
    target = torch.from_numpy(b["target"])
    loss = 2.0 * (ref_helpers.img2mse(tm, target) * 0.5 + ref_helpers.img2l1(tm, target) * 0.5)
    loss.backward()
    arrs = dict(meta=np.array([n_rand, Ns, 0, seed]), rgb_map=rgb.detach().numpy(), depth_map=depth.detach().numpy(),
                acc_map=acc.detach().numpy(), rgb_tm=tm.detach().numpy(), noise_rgb=noise.detach().numpy(), loss=np.array(loss.item()))
    arrs.update(pack_grads(_named_grads(net)))
    arrs["grad_none"] = np.array(sorted(set(canon_name(k) for k, p in net.named_parameters() if p.grad is None)))
    save("train_c1", **arrs)


def case_train_consist(seed=52, n_rand=12, Ns=64, Ni=64):
    """The COMBINED step after noisenerf_start_iter (run_lushnerf.py:625-661, round 5): the kernel-on training forward AND the
    aligned-pixel renders of the consistency branch, loss = image terms + 1e-2 * loss_rgb, ONE backward through both."""
    import random as pyrandom
    import contextlib, io
    V, ns, anchor = 5, 32, 3
    wts = synth.all_weights(NUM_IMG, seed, sharp=True, rbk_scale=2.0e4)
    net = build_ref(Ni, wts)
    net.train()
    b = synth.ray_batch(n_rand, seed, NUM_IMG)
    d = synth.draws(n_rand * 5, Ns, Ni, seed)
    rays = torch.from_numpy(b["rays"])
    kw = dict(perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False,
              near=0., far=1.)
    with ServeDraws([d["t_rand"], d["noise_c"], d["u"], d["noise_f"]]):
        out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info={"images_idx": torch.from_numpy(b["images_idx"])}, retraw=True,
                  force_naive=False, allkernel=False, kernel_pixel=torch.from_numpy(b["fq_mask"]).bool(), **kw)
    rgb_blur, rgb0_blur = out[0], out[1]
    poses = torch.from_numpy(synth.poses(V, seed))
    samples, ax, ay, cert_f, Align_Matrix, Align_Mask = _consist_tables(seed, V, ns, anchor)
    rk = dict(kw, perturb=False, raw_noise_std=0., inference=True, save_warped_ray_img=False)      # render_kwargs_test (:406-410)
    keep = (torch.Tensor.cuda, pyrandom.randint, np.random.randint)
    torch.Tensor.cuda = lambda self, *a, **k: self
    pyrandom.randint = lambda a, b: anchor
    np.random.randint = lambda lo, hi=None, size=None: samples.copy()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rgb_align, cert = net(H, W, K, 1 << 20, poses=poses, render_kwargs=rk, render_factor=0, rays_info=torch.arange(V),
                                  consist_loss=True, Align_matrix=Align_Matrix, Align_mask=Align_Mask)
    finally:
        torch.Tensor.cuda, pyrandom.randint, np.random.randint = keep
    mask = cert >= 0.8
    mean = ref_helpers.compute_mean_with_confidence(rgb_align, cert, 0.8)
    loss_rgb = torch.sum(torch.abs(rgb_align - mean.unsqueeze(0)) * mask.unsqueeze(2)) / len(mask[mask == 1])
    target = torch.from_numpy(b["target"])
    img = ref_helpers.img2mse(rgb_blur, target) * 0.5 + ref_helpers.img2l1(rgb_blur, target) * 0.5 \
        + ref_helpers.img2mse(rgb0_blur, target) * 0.5 + ref_helpers.img2l1(rgb0_blur, target) * 0.5
    loss = img + 1e-2 * loss_rgb               # run_lushnerf.py:658-659 (i > noisenerf_start_iter)
    loss.backward()
    arrs = dict(meta=np.array([n_rand, Ns, Ni, seed, V, ns, anchor]), samples=samples, ax=ax, ay=ay, cert_in=cert_f,
                rgb_blur=rgb_blur.detach().numpy(), rgb0_blur=rgb0_blur.detach().numpy(), rgb_align=rgb_align.detach().numpy(),
                certainty=cert.numpy(), loss=np.array(loss.item()), loss_img=np.array(img.item()), loss_rgb=np.array(loss_rgb.item()))
    arrs.update(pack_grads(_named_grads(net)))
    arrs["grad_none"] = np.array(sorted(set(canon_name(k) for k, p in net.named_parameters() if p.grad is None)))
    save("train_consist", **arrs)


NEW_CASES = {"sample_pdf_z": case_sample_pdf_z, "lindisp_white": case_lindisp_white, "eval_forward": case_eval_forward,
             "consistency": case_consistency, "trajectory": case_trajectory, "trajectory_long": case_trajectory_long,
             "train_c1": case_train_c1, "train_consist": case_train_consist}


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "checkpoint":
        case_checkpoint_layout()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in NEW_CASES:      # regenerate some of the later rounds' fixtures only
        for name in sys.argv[1:]:
            NEW_CASES[name]()
        for sub in ("", "models", "utils"):
            assert not os.path.isdir(os.path.join(REF, sub, "__pycache__")), sub
        sys.exit(0)
    case_checkpoint_layout()
    case_sample_pdf()
    case_rbk()
    case_render_rays("rays_c1_train", 64, 32, 0, True, False, 11)
    case_render_rays("rays_6464_train_sharp", 48, 64, 64, True, True, 12)
    case_render_rays("rays_6464_eval_sharp", 48, 64, 64, False, True, 13)
    case_train("train_naive_sharp", 32, 64, 64, True, True, 21)
    case_train("train_kernel_sharp", 12, 64, 64, False, True, 22)
    case_train("train_kernel_default", 12, 64, 64, False, False, 23, allkernel=False)
    for fn in NEW_CASES.values():
        fn()
    # the imported reference packages must be left untouched (gim/ ships its own
    # upstream __pycache__ dirs; we never import it)
    for sub in ("", "models", "utils"):
        assert not os.path.isdir(os.path.join(REF, sub, "__pycache__")), sub
