"""CPU oracle for the LuSh-NeRF ray-march hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU (fp32) restatement of the reference algorithm
(SURVEY.md section 8, rows a1-a17).  It is the checker for the HIP path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it.  The product (lush_nerf_amd/) never imports or falls back to it.

Parity status: PINNED.  tests/golden/*.npz hold inputs/outputs produced by the
real reference (imported from /root/reference in the build container by
tests/golden/make_golden.py); tests/test_oracle_golden.py checks this file
against them.

It is functional: every network is a dict {reference state_dict name -> tensor}
so the same weights can be loaded into the reference modules unchanged.  Random
draws are explicit arguments (``draws``), in the reference's draw order
(models/lushnerf.py:515, :322, utils/run_lushnerf_helpers.py:578, :322); when a
draw is None it is taken from torch's global generator with the same call shape
the reference uses, so a shared torch.manual_seed reproduces the reference.

All file:line citations are into /root/reference.

Every function also runs in float64 when its tensor inputs are float64 (the reference's explicit ``.float()`` casts
are then skipped): tests use that as the exact-arithmetic yardstick against which BOTH the fp32 oracle's and the
GPU's gradient errors are measured (tests/gpu_diag.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------- a1
def embed(x: Tensor, n_freqs: int) -> Tensor:
    """Positional encoding, utils/run_lushnerf_helpers.py:311-361.

    Layout is [x, sin(1x), cos(1x), sin(2x), cos(2x), ...]: frequency-major,
    then sin/cos, then xyz.  Frequencies are exactly 2**0 .. 2**(n_freqs-1).
    """
    outs = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------- a2/a3
def _lin(p: Params, name: str, x: Tensor) -> Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def nerf_mlp(p: Params, prefix: str, x: Tensor, in_ch: int, in_ch_views: int,
             depth: int, skips=(4,), return_alpha: bool = True) -> Tensor:
    """NeRF.forward (helpers:394-423) and NeRF_Noise.forward (helpers:483-512).

    ``depth`` trunk layers with ReLU; after layer index in ``skips`` the encoded
    point is concatenated IN FRONT of h.  alpha = Linear(h) (no activation),
    feature = Linear(h) (no activation), one views layer with ReLU, rgb head.
    NeRF returns [rgb, alpha]; NeRF_Noise returns rgb only (return_alpha=False).
    """
    pts, views = x[..., :in_ch], x[..., in_ch:in_ch + in_ch_views]
    h = pts
    for i in range(depth):
        h = F.relu(_lin(p, f"{prefix}.pts_linears.{i}", h))
        if i in skips:
            h = torch.cat([pts, h], -1)
    feature = _lin(p, f"{prefix}.feature_linear", h)
    hv = F.relu(_lin(p, f"{prefix}.views_linears.0", torch.cat([feature, views], -1)))
    rgb = _lin(p, f"{prefix}.rgb_linear", hv)
    if not return_alpha:
        return rgb
    alpha = _lin(p, f"{prefix}.alpha_linear", h)
    return torch.cat([rgb, alpha], -1)


# --------------------------------------------------------------------------- a4/a5
def mlpforward(p: Params, prefix: str, pts: Tensor, viewdirs: Tensor,
               multires: int = 10, multires_views: int = 4, depth: int = 8) -> Tensor:
    """NeRFAll.mlpforward, models/lushnerf.py:234-266 (netchunk slicing does not
    change results and is not restated)."""
    R, S, _ = pts.shape
    e = embed(pts.reshape(-1, 3), multires)
    d = embed(viewdirs[:, None].expand(R, S, 3).reshape(-1, 3), multires_views)
    out = nerf_mlp(p, prefix, torch.cat([e, d], -1), e.shape[-1], d.shape[-1], depth)
    return out.reshape(R, S, 4)


def mlpforward_noise(p: Params, prefix: str, pts_noise: Tensor, viewdirs: Tensor,
                     multires: int = 10, multires_views: int = 4, depth: int = 4) -> Tensor:
    """NeRFAll.mlpforward_noise, models/lushnerf.py:268-293: one point per ray,
    sample index 16 of the UN-jittered grid."""
    e = embed(pts_noise[:, 16], multires)
    d = embed(viewdirs, multires_views)
    return nerf_mlp(p, prefix, torch.cat([e, d], -1), e.shape[-1], d.shape[-1], depth,
                    return_alpha=False)


# --------------------------------------------------------------------------- a6
def raw2outputs(raw: Tensor, z_vals: Tensor, rays_d: Tensor, raw_noise_std: float = 0.,
                white_bkgd: bool = False, noise: Optional[Tensor] = None,
                training: bool = True, render_rmnearplane: float = 0.):
    """NeRFAll.raw2outputs, models/lushnerf.py:296-352.

    dists has S-1 entries (no 1e10 tail, :314-315); the last sample has
    alpha == 1 (:338); rgb = sigmoid (rgb_activate), density = relu
    (sigma_activate); in eval mode density is zeroed where z[1:] <=
    render_rmnearplane/128 (:331-335).  ``noise`` is the N(0,1) draw of shape
    [R, S-1] (:322) before scaling by raw_noise_std.
    """
    dists = (z_vals[..., 1:] - z_vals[..., :-1]) * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma_in = raw[..., :-1, 3]
    if raw_noise_std > 0.:
        if noise is None:
            noise = torch.randn_like(sigma_in)
        sigma_in = sigma_in + noise * raw_noise_std
    density = F.relu(sigma_in)
    if (not training) and render_rmnearplane > 0:
        density = (z_vals[:, 1:] > render_rmnearplane / 128).type_as(density) * density
    alpha = 1. - torch.exp(-density * dists)
    alpha = torch.cat([alpha, torch.ones_like(alpha[:, :1])], -1)
    # (1. + 1e-10) is a Python double that rounds to 1.0f when added to an fp32
    # tensor, so the reference's epsilon is a no-op (:341); keep the same form.
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]),
                                     -alpha + (1. + 1e-10)], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, density, acc_map, weights, depth_map


# --------------------------------------------------------------------------- a7
def sample_pdf(bins: Tensor, weights: Tensor, n_samples: int, det: bool,
               u: Optional[Tensor] = None) -> Tensor:
    """utils/run_lushnerf_helpers.py:566-609.  ``u`` is the U[0,1) draw
    [R, n_samples] (:578) when not det."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0., 1., steps=n_samples).to(cdf.dtype).expand(list(cdf.shape[:-1]) + [n_samples])
    elif u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous().to(cdf.dtype)
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


# --------------------------------------------------------------------------- a8/a9/a10
def _z_grid(near: Tensor, far: Tensor, n: int, lindisp: bool) -> Tensor:
    t = torch.linspace(0., 1., steps=n).to(near.dtype)
    if not lindisp:
        z = near * (1. - t) + far * t
    else:
        z = 1. / (1. / near * (1. - t) + 1. / far * t)
    return z.expand(near.shape[0], n)


def _stratify(z: Tensor, t_rand: Optional[Tensor]) -> Tensor:
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    if t_rand is None:
        t_rand = torch.rand(z.shape)
    return lower + (upper - lower) * t_rand


def render_rays(p: Params, ray_batch: Tensor, N_samples: int, retraw: bool = False,
                lindisp: bool = False, perturb: float = 0., N_importance: int = 0,
                white_bkgd: bool = False, raw_noise_std: float = 0.,
                draws: Optional[Dict[str, Tensor]] = None, with_noise_branch: bool = True,
                training: bool = True, render_rmnearplane: float = 0.,
                has_fine: bool = True):
    """NeRFAll.render_rays (models/lushnerf.py:354-479) when with_noise_branch,
    else NeRFAll.render_rays_nonoise (:481-583).  Returns (ret, ret_noise) or ret.

    draws keys (all optional): t_rand [R,Ns], noise_c [R,Ns-1], u [R,Ni],
    noise_f [R,Ns+Ni-1].
    """
    draws = draws or {}
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z_vals = _z_grid(near, far, N_samples, lindisp)
    pts_noise = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]
    if perturb > 0.:
        z_vals = _stratify(z_vals, draws.get("t_rand"))
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]
    raw = mlpforward(p, "mlp_coarse", pts, viewdirs)
    rgb_noise = None
    if with_noise_branch:
        rgb_noise = mlpforward_noise(p, "mlp_noise_coarse", pts_noise.detach(), viewdirs.detach())
    kw = dict(training=training, render_rmnearplane=render_rmnearplane)
    rgb_map, density, acc_map, weights, depth_map = raw2outputs(
        raw, z_vals, rays_d, raw_noise_std, white_bkgd, draws.get("noise_c"), **kw)
    ret = {}
    if N_importance > 0:
        rgb0, depth0, acc0, density0 = rgb_map, depth_map, acc_map, density
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.),
                               u=draws.get("u")).detach()
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]
        raw = mlpforward(p, "mlp_fine" if has_fine else "mlp_coarse", pts, viewdirs)
        rgb_map, density, acc_map, weights, depth_map = raw2outputs(
            raw, z_vals, rays_d, raw_noise_std, white_bkgd, draws.get("noise_f"), **kw)
        ret.update(rgb0=rgb0, depth0=depth0, acc0=acc0, density0=density0,
                   z_std=torch.std(z_samples, dim=-1, unbiased=False))
    ret.update(rgb_map=rgb_map, depth_map=depth_map, acc_map=acc_map, density_map=density)
    ret["_weights"] = weights   # oracle-only extras (not in the reference dict)
    ret["_z_vals"] = z_vals
    if retraw:
        ret["raw"] = raw
    if with_noise_branch:
        return ret, {"rgb_map": rgb_noise}
    return ret


def render_rays_noise(p: Params, ray_batch: Tensor, N_samples: int, lindisp: bool = False):
    """NeRFAll.render_rays_noise, models/lushnerf.py:585-617."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    z = _z_grid(ray_batch[:, 6:7], ray_batch[:, 7:8], N_samples, lindisp)
    pts_noise = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    return {"rgb_map": mlpforward_noise(p, "mlp_noise_coarse", pts_noise.detach(),
                                        ray_batch[:, -3:].detach())}


# --------------------------------------------------------------------------- a12
def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor):
    """utils/run_lushnerf_helpers.py:542-562."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    o0 = sx * o[..., 0] / o[..., 2]
    o1 = sy * o[..., 1] / o[..., 2]
    o2 = 1. + 2. * near / o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2. * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------- a11
def pack_rays(H: int, W: int, focal: float, rays: Tensor, ndc: bool = True,
              near: float = 0., far: float = 1.) -> Tensor:
    """Prologue shared by render_infer / render_train_scene / render_train_noise
    (models/lushnerf.py:706-729, 772-795, 827-850): rays[...,3,2] -> [R,11]
    = [o, d, near, far, viewdir] with viewdir normalised BEFORE the NDC map."""
    rays_o, rays_d = rays[..., 0], rays[..., 1]
    f32 = (lambda x: x) if rays.dtype == torch.float64 else (lambda x: x.float())   # .float() of :716, :727-729
    viewdirs = f32((rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3))
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rays_o, rays_d = f32(rays_o.reshape(-1, 3)), f32(rays_d.reshape(-1, 3))
    ones = torch.ones_like(rays_d[..., :1])
    return torch.cat([rays_o, rays_d, near * ones, far * ones, viewdirs], -1)


# --------------------------------------------------------------------------- a14
def se3_warp(pts: Tensor, rot: Tensor, trans: Tensor) -> Tensor:
    """SE3Field.warp + Rigid_body.exp_se3/exp_so3, utils/rigid_warping.py:20-140,
    in closed form (no 4x4 assembly): with theta=|rot|+1e-10, w=rot/theta,
    v=trans/theta, K=[w]x:
        R = I + sin(theta) K + (1-cos(theta)) K^2
        t = (theta I + (1-cos(theta)) K + (theta-sin(theta)) K^2) v
    """
    theta = torch.linalg.norm(rot, dim=-1, keepdim=True) + 1.0e-10
    w, v = rot / theta, trans / theta
    s, c = torch.sin(theta), torch.cos(theta)

    def cross(a, b):
        return torch.cross(a, b, dim=-1)

    wx_p = cross(w, pts)
    Rp = pts + s * wx_p + (1. - c) * cross(w, wx_p)
    wx_v = cross(w, v)
    t = theta * v + (1. - c) * wx_v + (theta - s) * cross(w, wx_v)
    return Rp + t


# --------------------------------------------------------------------------- a13
def rbk_forward(p: Params, rays: Tensor, images_idx: Tensor, num_motion: int = 4,
                rv_window: float = 0.1, depth: int = 4, prefix: str = "mlp_rbk"):
    """View_Embedding + Rigid_Blurring_Kernel.forward, models/lushnerf.py:27-35,
    118-153, with rbk_warp :75-98 (use_origin=True).  The trunk skip (index 4)
    never fires for depth 4.  r.reshape(N,3,M): motion i uses columns
    {i, M+i, 2M+i} of the 3M-wide head output (:76-77)."""
    e = p[f"{prefix}.view_embedding_layer.view_embed_layer.weight"][images_idx.reshape(-1)]
    h = e
    for i in range(depth):
        h = F.relu(_lin(p, f"{prefix}.view_embed_linears.{i}", h))
    h_r = F.relu(_lin(p, f"{prefix}.r_branch.0", h))
    h_v = F.relu(_lin(p, f"{prefix}.v_branch.0", h))
    h_w = F.relu(_lin(p, f"{prefix}.w_branch.0", h))
    r = _lin(p, f"{prefix}.r_linear", h_r) * rv_window
    v = _lin(p, f"{prefix}.v_linear", h_v) * rv_window
    w = torch.sigmoid(_lin(p, f"{prefix}.w_linear", h_w))
    w = w / (torch.sum(w, dim=-1, keepdim=True) + 1e-10)
    N = rays.shape[0]
    r, v = r.reshape(N, 3, num_motion), v.reshape(N, 3, num_motion)
    rays_o, rays_d = rays[..., 0], rays[..., 1]
    end = rays_o + rays_d
    out = [torch.stack([rays_o, rays_d], -1)]
    for i in range(num_motion):
        wo = se3_warp(rays_o, r[:, :, i], v[:, :, i])
        we = se3_warp(end, r[:, :, i], v[:, :, i])
        out.append(torch.stack([wo, we - wo], -1))
    new_rays = torch.stack(out, 1)                  # [N, M+1, 3, 2]
    return new_rays.reshape(-1, 3, 2), w


# --------------------------------------------------------------------------- a15
def rbk_weighted_sum(x: Tensor, ccw: Tensor) -> Tensor:
    """Rigid_Blurring_Kernel.rbk_weighted_sum, models/lushnerf.py:100-116, for one
    tensor of rank 1-3 whose leading dim is N*(M+1), ray-major motion-minor."""
    m = ccw.shape[1]
    xs = x.reshape(-1, m, *x.shape[1:])
    c = ccw.reshape(ccw.shape[0], m, *([1] * (x.dim() - 1)))
    return torch.sum(xs * c, dim=1)


def tonemap(x: Tensor, kind: str = "gamma") -> Tensor:
    """ToneMapping, utils/run_lushnerf_helpers.py:134-183 ('none' and 'gamma')."""
    return x if kind == "none" else x ** (1. / 2.2)


# --------------------------------------------------------------------------- a16
def forward_train(p: Params, H: int, W: int, focal: float, rays: Tensor, images_idx: Tensor,
                  N_samples: int, N_importance: int, force_naive: bool, perturb: float = 1.,
                  raw_noise_std: float = 1., allkernel: bool = False,
                  kernel_pixel: Optional[Tensor] = None, draws=None, tone: str = "gamma",
                  num_motion: int = 4, rv_window: float = 0.1, retraw: bool = True):
    """NeRFAll.forward training branch, models/lushnerf.py:630-662.

    Returns the reference 7-tuple; empty dicts are returned as {}.
    Blur-kernel branch (:636-654): RBK -> optional grad mask -> render_train_scene
    on N*(M+1) rays -> render_train_noise on the N input rays -> 0.1*sigmoid ->
    weighted sum -> tone map.  Naive branch (:657-662): render_infer; the noise is
    returned but NOT added to the colours.
    """
    kw = dict(N_samples=N_samples, N_importance=N_importance, perturb=perturb,
              raw_noise_std=raw_noise_std, retraw=retraw)
    if not force_naive:
        rays_t, ccw = rbk_forward(p, rays, images_idx, num_motion, rv_window)
        if allkernel:
            m = kernel_pixel.reshape(-1).bool().repeat_interleave(num_motion + 1)
            rays_t = torch.where(m[:, None, None], rays_t, rays_t.detach())
        batch = pack_rays(H, W, focal, rays_t)
        ret = render_rays(p, batch, draws=draws, with_noise_branch=False, **kw)
        nbatch = pack_rays(H, W, focal, rays)
        rgb_noise = 0.1 * torch.sigmoid(render_rays_noise(p, nbatch, N_samples)["rgb_map"])
        rgb = rbk_weighted_sum(ret["rgb_map"], ccw)
        rgb0 = rbk_weighted_sum(ret["rgb0"], ccw)
        return (tonemap(rgb + rgb_noise, tone), tonemap(rgb0 + rgb_noise, tone), {},
                rgb_noise, rgb_noise, tonemap(rgb, tone), tonemap(rgb0, tone))
    batch = pack_rays(H, W, focal, rays)
    ret, ret_noise = render_rays(p, batch, draws=draws, with_noise_branch=True, **kw)
    rgb_noise = 0.1 * torch.sigmoid(ret_noise["rgb_map"])
    return (tonemap(ret["rgb_map"], tone), tonemap(ret["rgb0"], tone), {}, rgb_noise,
            rgb_noise, {}, {})


# --------------------------------------------------------------------------- a17
def train_loss(rgb_blur: Tensor, rgb0_blur: Tensor, target: Tensor) -> Tensor:
    """run_lushnerf.py:652-661 (i <= noisenerf_start_iter): 0.5*MSE + 0.5*L1 on the
    fine and on the coarse colour."""
    def half(x):
        return 0.5 * torch.mean((x - target) ** 2) + 0.5 * torch.mean(torch.abs(x - target))
    return half(rgb_blur) + half(rgb0_blur)


def lr_at(step: int, lrate: float = 5e-4, lrate_decay: int = 250) -> float:
    """run_lushnerf.py:681-685."""
    return lrate * (0.1 ** (step / (lrate_decay * 1000)))


def get_rays_np_formula(H: int, W: int, focal: float, c2w, px, py):
    """utils/run_lushnerf_helpers.py:531-539 evaluated at pixel (px, py) (float
    arrays): dirs = [(i+0.5-cx)/f, -(j+0.5-cy)/f, -1]; rays_d = R dirs; rays_o = t."""
    import numpy as np
    dirs = np.stack([(px + (0.5 - W / 2)) / focal, -(py + (0.5 - H / 2)) / focal,
                     -np.ones_like(px)], -1)
    rays_d = np.sum(dirs[..., None, :] * c2w[..., :3, :3], -1)
    rays_o = np.broadcast_to(c2w[..., :3, 3], rays_d.shape)
    return rays_o.astype(np.float32), rays_d.astype(np.float32)


def lr_schedule(n_steps: int, lrate: float = 5e-4, lrate_decay: int = 250, start: int = 0):
    """Learning rate each optimizer.step() of the reference loop actually runs with: the rate is recomputed from
    global_step AFTER the step and BEFORE global_step += 1 (run_lushnerf.py:675-685, 788), so step g uses the
    rate of global_step g-1 and the first step the constructor's lrate (:368-371)."""
    used, lr, global_step = [], lrate, start
    for _ in range(n_steps):
        used.append(lr)                                   # optimizer.step()
        lr = lr_at(global_step, lrate, lrate_decay)       # new_lrate -> param_group['lr']
        global_step += 1
    return used


# --------------------------------------------------------------------------- f3: consistency branch
def get_rays(H: int, W: int, K, c2w: Tensor):
    """utils/run_lushnerf_helpers.py:517-528 (pixel centres: HALF_PIX = 0.5)."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i + (0.5 - K[0][2])) / K[0][0], -(j + (0.5 - K[1][2])) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def render_aligned_pixel(p: Params, H: int, W: int, focal: float, poses: Tensor, align_anchor: Tensor,
                         cert_anchor: Tensor, samples: Tensor, N_samples: int, N_importance: int):
    """NeRFAll.Render_Aligned_Pixel, models/lushnerf.py:949-989, after the two host draws (:960, :964):
    ``align_anchor`` = Align_matrix[anchor] [V, HW, 4], ``cert_anchor`` = Align_mask[anchor] [V, HW].
    Module in train mode, render_kwargs_test (perturb False, raw_noise_std 0): no draws, no near-plane mask.
    Returns (rgb_align [V, ns, 3], certainty [V, ns])."""
    K = [[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]]
    anchor_pose = align_anchor[:, samples]
    anchor_certain = cert_anchor[:, samples]
    V, ns = poses.shape[0], samples.numel()
    rgb_align = torch.zeros(V, ns, 3)
    cert = torch.zeros(V, ns)
    outs = []
    for i in range(V):
        ro, rd = get_rays(H, W, K, poses[i])
        rays_org = torch.stack([ro, rd], -1)                      # [H, W, 3, 2]
        px = anchor_pose[i, :, 2:].long()
        rays = rays_org[torch.clamp(px[:, 1], 0, H - 1), torch.clamp(px[:, 0], 0, W - 1)]
        batch = pack_rays(H, W, focal, rays.detach())
        ret = render_rays(p, batch, N_samples, perturb=0., N_importance=N_importance, raw_noise_std=0.,
                          with_noise_branch=False, training=True)
        outs.append(ret["rgb_map"])
        cert[i] = anchor_certain[i].float()
    return torch.stack(outs, 0), cert


def compute_mean_with_confidence(rgb_align: Tensor, confidence: Tensor, threshold: float = 0.2) -> Tensor:
    """utils/run_lushnerf_helpers.py:665-688 (vectorised; the per-sample in-place adds are differentiable)."""
    m = (confidence >= threshold).to(rgb_align.dtype)                 # [V, ns]
    cnt = m.sum(0)
    mean = (rgb_align * m[..., None]).sum(0)
    return mean / torch.where(cnt == 0, torch.ones_like(cnt), cnt)[:, None]


def consist_loss(rgb_align: Tensor, certainty: Tensor, threshold: float = 0.8) -> Tensor:
    """run_lushnerf.py:644-650: masked L1 around the confidence-weighted mean."""
    mask = certainty >= threshold
    mean = compute_mean_with_confidence(rgb_align, certainty, threshold)
    return torch.sum(torch.abs(rgb_align - mean[None]) * mask[..., None]) / mask.sum()
