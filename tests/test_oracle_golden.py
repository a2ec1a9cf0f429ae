"""The oracle (oracle/lush_oracle.py) against the reference-generated fixtures.

Tolerances: the oracle and the reference are both fp32 torch-CPU; they differ only
in op order (closed-form SE(3), fused gathers), so 2e-5 normalised max error.
"""
import numpy as np
import pytest
import torch

from lush_nerf_amd import synth
from oracle import lush_oracle as O
from tests import util

TOL = 2e-5


def _batch(n, seed):
    b = synth.ray_batch(n, seed, util.NUM_IMG)
    return {k: torch.from_numpy(v) for k, v in b.items()}


def test_sample_pdf_golden():
    g = util.golden("sample_pdf")
    bins, w, u = (torch.from_numpy(g[k]) for k in ("bins", "weights", "u"))
    assert util.relerr(O.sample_pdf(bins, w, 64, det=False, u=u), g["s_rand"]) < 1e-6
    assert util.relerr(O.sample_pdf(bins, w, 64, det=True), g["s_det"]) < 1e-6


def test_rbk_and_ndc_golden():
    g = util.golden("rbk")
    n, seed = (int(x) for x in g["meta"])
    p = util.params(seed, rbk_scale=3.0e5)
    b = _batch(n, seed)
    new_rays, ccw = O.rbk_forward(p, b["rays"], b["images_idx"])
    assert util.relerr(new_rays, g["new_rays"]) < TOL
    assert util.relerr(ccw, g["ccw"]) < TOL
    o, d = O.ndc_rays(util.H, util.W, util.FOCAL, 1., new_rays[..., 0], new_rays[..., 1])
    assert util.relerr(o, g["ndc_o"]) < TOL and util.relerr(d, g["ndc_d"]) < TOL
    # the warp must not be the identity in this fixture
    assert float((new_rays.reshape(n, 5, 3, 2)[:, 1:] - new_rays.reshape(n, 5, 3, 2)[:, :1]).abs().max()) > 1e-2


@pytest.mark.parametrize("name", ["rays_c1_train", "rays_6464_train_sharp", "rays_6464_eval_sharp"])
def test_render_rays_golden(name):
    g = util.golden(name)
    n, Ns, Ni, train, sharp, seed = (int(x) for x in g["meta"])
    p = util.params(seed, sharp=bool(sharp))
    b = _batch(n, seed)
    batch = O.pack_rays(util.H, util.W, util.FOCAL, b["rays"])
    with torch.no_grad():
        ret, ret_noise = O.render_rays(
            p, batch, Ns, retraw=True, perturb=1. if train else 0., N_importance=Ni,
            raw_noise_std=1. if train else 0., draws=util.tdraws(n, Ns, Ni, seed) if train else None,
            training=bool(train), render_rmnearplane=80)
    keys = ["rgb_map", "depth_map", "acc_map", "density_map", "raw"]
    if Ni > 0:
        keys += ["rgb0", "depth0", "acc0", "density0", "z_std"]
    for k in keys:
        assert util.relerr(ret[k], g[k]) < TOL, k
    assert util.relerr(ret_noise["rgb_map"], g["noise_rgb"]) < TOL
    if sharp:   # the fixture must exercise the compositing scan, not only the last sample
        assert float(ret["_weights"][:, -1].mean()) < 0.5


@pytest.mark.parametrize("name", ["train_naive_sharp", "train_kernel_sharp", "train_kernel_default"])
def test_forward_train_golden(name):
    g = util.golden(name)
    n, Ns, Ni, naive, sharp, seed, allk = (int(x) for x in g["meta"])
    p = util.params(seed, sharp=bool(sharp), rbk_scale=1.0 if naive else 2.0e4, requires_grad=True)
    b = _batch(n, seed)
    rays = b["rays"].clone().requires_grad_(True)
    out = O.forward_train(p, util.H, util.W, util.FOCAL, rays, b["images_idx"], Ns, Ni,
                          force_naive=bool(naive), allkernel=bool(allk), kernel_pixel=b["fq_mask"],
                          draws=util.tdraws(n * (1 if naive else 5), Ns, Ni, seed))
    loss = O.train_loss(out[0], out[1], b["target"])
    loss.backward()
    assert util.relerr(out[0], g["rgb_blur"]) < TOL
    assert util.relerr(out[1], g["rgb0_blur"]) < TOL
    assert util.relerr(out[3], g["noise"]) < TOL
    if not naive:
        assert util.relerr(out[5], g["rgb"]) < TOL and util.relerr(out[6], g["rgb0"]) < TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    none = set(str(x) for x in g["grad_none"])
    for k, v in p.items():
        assert (v.grad is None) == (k in none), k
    # Gradient gate.  Naive case: forward is bit-identical to the reference, so 2e-4.
    # Kernel-on case: the closed-form SE(3) warp rounds differently from the
    # reference's 4x4 matmul, rays move by ~1 ulp, a few of the 25M ReLU kinks
    # flip and gradient terms change discretely.  Measured: the oracle's OWN
    # gradients move by 5.5e-3 (L2) / 1.0e-2 (max) under a 1-ulp ray perturbation
    # while outputs move 8e-8 (DESIGN.md "gradient conditioning").  Gate = 3e-2.
    gtol = 2e-4 if naive else 3e-2
    util.check_grads({k: v.grad for k, v in p.items()}, g, gtol)
    if g["grad_rays"].size:
        assert util.relerr(rays.grad, g["grad_rays"]) < gtol


# ----------------------------------------------------------------------------- round-2 fixtures
def test_sample_pdf_on_midpoints_golden():
    """bins = mid-points of a stored z: the vectors the GPU kernel (which takes z) is also run on."""
    g = util.golden("sample_pdf_z")
    z, w, u = (torch.from_numpy(g[k]) for k in ("z", "weights", "u"))
    mid = .5 * (z[:, 1:] + z[:, :-1])
    s_rand = O.sample_pdf(mid, w[:, 1:-1], 64, det=False, u=u)
    s_det = O.sample_pdf(mid, w[:, 1:-1], 64, det=True)
    assert util.relerr(s_rand, g["s_rand"]) < 1e-6 and util.relerr(s_det, g["s_det"]) < 1e-6
    assert util.relerr(torch.sort(torch.cat([z, s_rand], -1), -1)[0], g["merged_rand"]) < 1e-6


def nondc_batch(n, seed):
    b = synth.ray_batch(n, seed, util.NUM_IMG)
    o, d = b["rays"][..., 0], b["rays"][..., 1]
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    ones = np.ones((n, 1), np.float32)
    return torch.from_numpy(np.concatenate([o, d, 2.0 * ones, 6.0 * ones, vd], -1).astype(np.float32))


def test_lindisp_white_bkgd_golden():
    """lindisp=True (sampling linear in disparity, models/lushnerf.py:393-396) and white_bkgd=True (:349-350)."""
    g = util.golden("rays_lindisp_white")
    n, Ns, Ni, seed = (int(x) for x in g["meta"])
    p = util.params(seed, sharp=True)
    with torch.no_grad():
        ret, ret_noise = O.render_rays(p, nondc_batch(n, seed), Ns, retraw=True, lindisp=True, perturb=1.,
                                       N_importance=Ni, white_bkgd=True, raw_noise_std=1.,
                                       draws=util.tdraws(n, Ns, Ni, seed))
    for k in ("rgb_map", "depth_map", "acc_map", "density_map", "raw", "rgb0", "depth0", "acc0", "z_std"):
        assert util.relerr(ret[k], g[k]) < TOL, k
    assert util.relerr(ret_noise["rgb_map"], g["noise_rgb"]) < TOL
    assert float(ret["_z_vals"].min()) >= 2.0 - 1e-5 and float(ret["_z_vals"].max()) <= 6.0 + 1e-5


def test_eval_forward_golden():
    """NeRFAll.forward(poses=...) in eval mode: render_path over all pixels, near-plane mask, tone map."""
    g = util.golden("eval_forward")
    H, W, seed = (int(x) for x in g["meta"])
    F = float(g["focal"])
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    p = util.params(seed, sharp=True)
    poses = torch.from_numpy(synth.poses(2, seed))
    rgbs, noises, depths = [], [], []
    with torch.no_grad():
        for c2w in poses:
            ro, rd = O.get_rays(H, W, K, c2w)
            batch = O.pack_rays(H, W, F, torch.stack([ro, rd], -1))
            ret, rn = O.render_rays(p, batch, 64, perturb=0., N_importance=64, raw_noise_std=0., training=False,
                                    render_rmnearplane=80)
            rgbs.append(ret["rgb_map"].reshape(H, W, 3))
            depths.append(ret["depth_map"].reshape(H, W))
            noises.append(rn["rgb_map"].reshape(H, W, 3))
    assert util.relerr(O.tonemap(torch.stack(rgbs)), g["rgbs"]) < TOL
    assert util.relerr(O.tonemap(0.1 * torch.sigmoid(torch.stack(noises))), g["noise"]) < TOL
    assert util.relerr(torch.stack(depths), g["depths"]) < 5e-5


def consistency_inputs(g):
    """Dense Align_matrix[anchor] / Align_mask[anchor] rebuilt from the sparse fixture (values only exist at the samples)."""
    V, ns, seed, anchor = (int(x) for x in g["meta"])
    HW = util.H * util.W
    st = torch.from_numpy(g["samples"])
    am = torch.zeros(V, HW, 4)
    am[:, st, 2] = torch.from_numpy(g["ax"])
    am[:, st, 3] = torch.from_numpy(g["ay"])
    cm = torch.zeros(V, HW, dtype=torch.bool)
    cm[:, st] = torch.from_numpy(g["cert_in"]) != 0
    return V, ns, seed, anchor, st, am, cm


def test_consistency_branch_golden():
    """Render_Aligned_Pixel + compute_mean_with_confidence + masked L1, with gradients."""
    g = util.golden("consistency")
    V, ns, seed, anchor, st, am, cm = consistency_inputs(g)
    p = util.params(seed, sharp=True, requires_grad=True)
    poses = torch.from_numpy(synth.poses(V, seed))
    rgb_align, cert = O.render_aligned_pixel(p, util.H, util.W, util.FOCAL, poses, am, cm, st, 64, 64)
    assert util.relerr(rgb_align, g["rgb_align"]) < TOL
    assert np.array_equal(cert.numpy(), g["certainty"])
    assert util.relerr(O.compute_mean_with_confidence(rgb_align, cert, 0.8), g["mean"]) < TOL
    loss = O.consist_loss(rgb_align, cert, 0.8)
    assert abs(loss.item() - float(g["loss_rgb"])) < 1e-6
    loss.backward()
    none = set(str(x) for x in g["grad_none"])
    for k, v in p.items():
        assert (v.grad is None) == (k in none), k
    util.check_grads({k: v.grad for k, v in p.items()}, g, 2e-4)   # forward is bit-compatible: no kink flips


# ----------------------------------------------------------------------------- round-5 fixtures
def oracle_c1_step(p, rays, target, Ns, draws):
    """BASELINE config 1 as a training step (tests/golden/make_golden.py case_train_c1): render_infer -> render_rays at
    N_importance = 0 (models/lushnerf.py:679-763, :354-479), the loss of run_lushnerf.py:652-661 with rgb0 = rgb."""
    batch = O.pack_rays(util.H, util.W, util.FOCAL, rays)
    ret, ret_noise = O.render_rays(p, batch, Ns, retraw=True, perturb=1., N_importance=0, raw_noise_std=1., draws=draws)
    tm = O.tonemap(ret["rgb_map"])
    return ret, ret_noise, tm, O.train_loss(tm, tm, target)


def test_c1_training_step_golden():
    g = util.golden("train_c1")
    n, Ns, Ni, seed = (int(x) for x in g["meta"])
    p = {k: v for k, v in util.params(seed, sharp=True, requires_grad=True).items() if not k.startswith("mlp_fine.")}
    b = _batch(n, seed)
    ret, ret_noise, tm, loss = oracle_c1_step(p, b["rays"], b["target"], Ns, util.tdraws(n, Ns, 0, seed))
    loss.backward()
    for k, v in (("rgb_map", ret["rgb_map"]), ("depth_map", ret["depth_map"]), ("acc_map", ret["acc_map"]), ("rgb_tm", tm),
                 ("noise_rgb", ret_noise["rgb_map"])):
        assert util.relerr(v, g[k]) < TOL, k
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    none = set(str(x) for x in g["grad_none"])
    for k, v in p.items():
        assert (v.grad is None) == (k in none), k
    assert any(v.grad is not None for v in p.values())
    util.check_grads({k: v.grad for k, v in p.items()}, g, 2e-4)      # naive: the forward is bit-compatible, no kink flips


def consist_step_inputs(g):
    n, Ns, Ni, seed, V, ns, anchor = (int(x) for x in g["meta"])
    HW = util.H * util.W
    st = torch.from_numpy(g["samples"])
    am = torch.zeros(V, HW, 4)
    am[:, st, 2] = torch.from_numpy(g["ax"])
    am[:, st, 3] = torch.from_numpy(g["ay"])
    cm = torch.zeros(V, HW, dtype=torch.bool)
    cm[:, st] = torch.from_numpy(g["cert_in"]) != 0
    return n, Ns, Ni, seed, V, ns, anchor, st, am, cm


def oracle_consist_step(p, b, Ns, Ni, draws, poses, am, cm, st, weight=1e-2, dt=torch.float32):
    """The combined step of run_lushnerf.py:625-661 for i > noisenerf_start_iter: image terms of the kernel-on forward plus
    weight x the masked L1 of the aligned-pixel renders; returns (outputs, rgb_align, loss_img, loss_rgb, loss)."""
    out = O.forward_train(p, util.H, util.W, util.FOCAL, b["rays"].to(dt), b["images_idx"], Ns, Ni, force_naive=False,
                          allkernel=False, kernel_pixel=b["fq_mask"], draws=draws)
    img = O.train_loss(out[0], out[1], b["target"].to(dt))
    ra, ca = O.render_aligned_pixel(p, util.H, util.W, util.FOCAL, poses.to(dt), am.to(dt), cm, st, Ns, Ni)
    lrgb = O.consist_loss(ra, ca.to(dt), 0.8)
    return out, ra, img, lrgb, img + weight * lrgb


def test_combined_consistency_step_golden():
    """One backward through BOTH the ray batch and the aligned-pixel renders (loss += 1e-2 * loss_rgb)."""
    g = util.golden("train_consist")
    n, Ns, Ni, seed, V, ns, anchor, st, am, cm = consist_step_inputs(g)
    p = util.params(seed, sharp=True, rbk_scale=2.0e4, requires_grad=True)
    b = _batch(n, seed)
    poses = torch.from_numpy(synth.poses(V, seed))
    out, ra, img, lrgb, loss = oracle_consist_step(p, b, Ns, Ni, util.tdraws(n * 5, Ns, Ni, seed), poses, am, cm, st)
    loss.backward()
    assert util.relerr(out[0], g["rgb_blur"]) < TOL and util.relerr(out[1], g["rgb0_blur"]) < TOL
    assert util.relerr(ra, g["rgb_align"]) < TOL
    assert abs(img.item() - float(g["loss_img"])) < 1e-6 and abs(lrgb.item() - float(g["loss_rgb"])) < 1e-6
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    none = set(str(x) for x in g["grad_none"])
    for k, v in p.items():
        assert (v.grad is None) == (k in none), k
    util.check_grads({k: v.grad for k, v in p.items()}, g, 3e-2)      # kernel on: kink flips behind the closed-form warp (see above)


def test_lr_schedule_matches_reference_loop():
    """run_lushnerf.py:675-685, 788: the rate is updated after optimizer.step() from the un-incremented global_step."""
    used = O.lr_schedule(5)
    assert used[0] == used[1] == 5e-4
    assert used[2] == O.lr_at(1) and used[4] == O.lr_at(3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_clamped_samples_have_exactly_zero_gradient_rows(dtype):
    """What the live-point backward rests on (include/lush_march.h "Live points"), checked on the restatement of the reference's own
    raw2outputs (models/lushnerf.py:296-352) under autograd: a sample whose density pre-activation + noise is <= 0 (ReLU clamped,
    :313) and that is not the last of its ray (alpha = 1, :338) receives an EXACTLY zero gradient in all four of its raw values,
    whatever flows into the five outputs -- and every other sample a non-zero one."""
    g = torch.Generator().manual_seed(3)
    R, S = 64, 48
    raw = (torch.randn(R, S, 4, generator=g, dtype=dtype) * 1.5).requires_grad_(True)
    z = torch.sort(torch.rand(R, S, generator=g, dtype=dtype), -1)[0]
    d = torch.randn(R, 3, generator=g, dtype=dtype)
    noise = torch.randn(R, S - 1, generator=g, dtype=dtype)
    for white in (False, True):
        raw.grad = None
        rgb_map, density, acc, weights, depth = O.raw2outputs(raw, z, d, raw_noise_std=1.0, white_bkgd=white, noise=noise)
        ups = [torch.randn(t.shape, generator=g, dtype=dtype) for t in (rgb_map, density, acc, weights, depth)]
        torch.autograd.backward([rgb_map, density, acc, weights, depth], ups)
        dead = torch.zeros(R, S, dtype=torch.bool)
        dead[:, :-1] = (raw.detach()[:, :-1, 3] + noise) <= 0
        assert 0.3 < float(dead.float().mean()) < 0.7
        assert float(raw.grad[dead].abs().max()) == 0.0
        assert bool((raw.grad[~dead].abs().sum(-1) > 0).all())


def test_trained_like_field_has_empty_space_and_surfaces():
    """bench.py's `trained_like` workload (synth.all_weights(trained_like=True)): the density field it starts from has what the
    initialisation's does not -- empty space that is dead under the unit noise, and surfaces far above it -- and the fine network
    agrees with the coarse one, so sample_pdf's new samples land where the fine density is: under the oracle, on the synthetic rays,
    the coarse pass has roughly a tenth of its samples live, the fine pass more (the importance samples), far fewer than the ~0.45
    of the initialisation."""
    import torch
    from lush_nerf_amd import synth
    from oracle import lush_oracle as O
    b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(96, 1000, 30, step=0).items()}
    d = {k: torch.from_numpy(v) for k, v in synth.draws(96, 64, 64, 0, step=0).items()}
    batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, b["rays"])
    shares = {}
    for name, tl in (("init", False), ("trained_like", True)):
        w = synth.all_weights(30, 0, trained_like=tl)
        if tl:
            assert all((w["mlp_fine." + k[11:]] == v).all() for k, v in w.items() if k.startswith("mlp_coarse."))
        p = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
        with torch.no_grad():
            ret, _ = O.render_rays(p, batch, 64, retraw=True, perturb=1., N_importance=64, raw_noise_std=1., draws=d)
        sig = ret["raw"][..., 3]
        live = ((sig[:, :-1] + d["noise_f"]) > 0).float().mean().item()
        shares[name] = (live, float((sig > 1.0).float().mean()), float((sig < -1.0).float().mean()))
    assert 0.35 < shares["init"][0] < 0.6 and shares["init"][1] == 0.0, shares              # sigma ~ 0: the noise alone decides
    live, above, below = shares["trained_like"]
    assert 0.15 < live < 0.45 and above > 0.12 and below > 0.5, shares                      # surfaces, and mostly empty space
