"""Build-owned deterministic generator for weights and LLFF-shaped ray batches.

Everything here is derived from splitmix64, so the same seed gives the same
numbers on any box and any NumPy version.  Used by the
tests, the golden-fixture script, bench.py and smoke(); numpy only.

Weight names and shapes are the reference's state_dict names
(models/lushnerf.py:197-220, utils/run_lushnerf_helpers.py:379-390) so a dict
made here loads into the reference modules unchanged.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1), 53-bit, reproducible."""
    idx = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
    return (_splitmix64(idx, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(shape, lo: float, hi: float, seed: int, stream: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(n, seed, stream)).reshape(shape).astype(np.float32)


def normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(2 * n, seed, stream)
    r = np.sqrt(-2.0 * np.log(1.0 - u[:n]))
    return (r * np.cos(2.0 * math.pi * u[n:])).reshape(shape).astype(np.float32)


def _stream(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFF
    return h


def _linear(out: Dict[str, np.ndarray], name: str, fan_out: int, fan_in: int, seed: int,
            bound=None):
    b = 1.0 / math.sqrt(fan_in) if bound is None else bound
    out[name + ".weight"] = uniform((fan_out, fan_in), -b, b, seed, _stream(name + ".w"))
    bb = 1.0 / math.sqrt(fan_in)
    out[name + ".bias"] = uniform((fan_out,), -bb, bb, seed, _stream(name + ".b"))


def nerf_weights(prefix: str, D: int, W: int, seed: int, in_ch: int = 63, in_views: int = 27,
                 skips=(4,)) -> Dict[str, np.ndarray]:
    """NeRF / NeRF_Noise parameter set (helpers:379-390), nn.Linear-equivalent
    U(+-1/sqrt(fan_in)) init."""
    p: Dict[str, np.ndarray] = {}
    for i in range(D):
        fan_in = in_ch if i == 0 else (W + in_ch if (i - 1) in skips else W)
        _linear(p, f"{prefix}.pts_linears.{i}", W, fan_in, seed)
    _linear(p, f"{prefix}.views_linears.0", W // 2, in_views + W, seed)
    _linear(p, f"{prefix}.feature_linear", W, W, seed)
    _linear(p, f"{prefix}.alpha_linear", 1, W, seed)
    _linear(p, f"{prefix}.rgb_linear", 3, W // 2, seed)
    return p


def rbk_weights(num_img: int, seed: int, embed_ch: int = 64, W: int = 64, D: int = 4,
                W_b: int = 32, num_motion: int = 4, prefix: str = "mlp_rbk") -> Dict[str, np.ndarray]:
    """Rigid_Blurring_Kernel + View_Embedding parameters (models/lushnerf.py:38-73)."""
    p: Dict[str, np.ndarray] = {}
    p[f"{prefix}.view_embedding_layer.view_embed_layer.weight"] = normal(
        (num_img, embed_ch), seed, _stream("embed"))
    for i in range(D):
        _linear(p, f"{prefix}.view_embed_linears.{i}", W, embed_ch if i == 0 else W, seed)
    tiny = 1e-5 * 6.0 / (W_b + 3 * num_motion)       # xavier_uniform with the reference gain
    _linear(p, f"{prefix}.r_branch.0", W_b, W, seed)
    _linear(p, f"{prefix}.r_linear", 3 * num_motion, W_b, seed, bound=tiny)
    _linear(p, f"{prefix}.v_branch.0", W_b, W, seed)
    _linear(p, f"{prefix}.v_linear", 3 * num_motion, W_b, seed, bound=tiny)
    _linear(p, f"{prefix}.w_branch.0", W_b, W, seed)
    _linear(p, f"{prefix}.w_linear", num_motion + 1, W_b, seed)
    return p


# ``trained_like``: alpha_linear x TRAINED_SCALE, bias TRAINED_BIAS (see all_weights).  Calibrated for seed 0 on the synthetic rays'
# points: the coarse network's w_alpha . h_7 there is -0.0168 +- 0.0089, so sigma = 3000 w.h + 20 = -30 +- 27: positive (mean +13,
# alpha ~ 0.3 per sample: a surface takes a handful of samples) in ~13 % of the volume, below -1 (dead under the unit noise) in ~85 %; measured on the GPU (tools/trained_like_probe.py): coarse pass 0.106 live, fine pass 0.289.
TRAINED_SCALE, TRAINED_BIAS = 3000.0, 20.0


def all_weights(num_img: int = 30, seed: int = 0, netwidth: int = 256, netdepth: int = 8,
                sharp: bool = False, rbk_scale: float = 1.0, trained_like=False) -> Dict[str, np.ndarray]:
    """Full NeRFAll parameter set.  ``sharp`` rescales alpha_linear so raw sigma spans
    roughly 0..100 (default init leaves the compositing scan and
    sample_pdf nearly untested).  ``rbk_scale`` multiplies r_linear/v_linear weights
    so the SE(3) warp is not numerically the identity in tests.
    ``trained_like`` (True, or a (scale, bias) pair): a density field shaped like a trained scene's instead of the
    initialisation's sigma = -0.003 +- 0.008 everywhere -- alpha_linear x scale with a bias chosen so that raw sigma is
    strongly negative in most of the volume (dead even under raw_noise_std = 1, configs/*_lushnerf:21) and strongly positive
    (tens, far above the noise) in the blobs where the random field peaks -- its "surfaces"; and the fine network is a COPY of
    the coarse one, as a trained pair agrees on where the surfaces are (so sample_pdf's new samples land where the fine
    network's density is, too).  What bench.py's ``trained_like`` workload starts from."""
    p: Dict[str, np.ndarray] = {}
    p.update(nerf_weights("mlp_coarse", netdepth, netwidth, seed + 1))
    p.update(nerf_weights("mlp_fine", netdepth, netwidth, seed + 2))
    p.update(nerf_weights("mlp_noise_coarse", netdepth // 2, netwidth // 2, seed + 3))
    p.update(rbk_weights(num_img, seed + 4))
    if sharp:
        for net in ("mlp_coarse", "mlp_fine"):
            p[f"{net}.alpha_linear.weight"] = p[f"{net}.alpha_linear.weight"] * 400.0
            p[f"{net}.alpha_linear.bias"] = p[f"{net}.alpha_linear.bias"] * 0.0 + 20.0
    if trained_like:
        scale, bias = (TRAINED_SCALE, TRAINED_BIAS) if trained_like is True else trained_like
        for k in [k for k in p if k.startswith("mlp_coarse.")]:
            p["mlp_fine." + k[len("mlp_coarse."):]] = p[k].copy()
        for net in ("mlp_coarse", "mlp_fine"):
            p[f"{net}.alpha_linear.weight"] = p[f"{net}.alpha_linear.weight"] * np.float32(scale)
            p[f"{net}.alpha_linear.bias"] = p[f"{net}.alpha_linear.bias"] * np.float32(0.0) + np.float32(bias)
    if rbk_scale != 1.0:
        for n in ("r_linear", "v_linear"):
            p[f"mlp_rbk.{n}.weight"] = p[f"mlp_rbk.{n}.weight"] * rbk_scale
    return p


# ----------------------------------------------------------------------------- rays
H_DEF, W_DEF, FOCAL_DEF = 640, 1120, 1000.0


def poses(num_img: int = 30, seed: int = 0) -> np.ndarray:
    """[num_img,3,4] forward-facing cameras: identity perturbed by <=5 degree
    rotations about x,y,z and U(-0.3,0.3)^3 translations."""
    ang = uniform((num_img, 3), -5.0, 5.0, seed, _stream("ang")).astype(np.float64) * math.pi / 180
    tr = uniform((num_img, 3), -0.3, 0.3, seed, _stream("tr")).astype(np.float64)
    out = np.zeros((num_img, 3, 4))
    for i in range(num_img):
        cx, cy, cz = np.cos(ang[i])
        sx, sy, sz = np.sin(ang[i])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        out[i, :, :3] = Rz @ Ry @ Rx
        out[i, :, 3] = tr[i]
    return out.astype(np.float32)


def pixel_batch(n_rand: int, seed: int = 0, num_img: int = 30, H: int = H_DEF, W: int = W_DEF, step: int = 0):
    """The (view, pixel) draws of one training batch + targets, without the rays: what a trainer that generates its
    rays on the device (lush_gen_rays) keeps resident instead of the [N_img*H*W, 2, 3] ray table.
    view, px, py [N] int64; target [N,3]; fq_mask [N] uint8."""
    st = _stream("batch") + 7919 * step
    views = np.minimum((uniform01(n_rand, seed, st) * num_img).astype(np.int64), num_img - 1)
    px = np.floor(uniform01(n_rand, seed, st + 1) * W).astype(np.int64)
    py = np.floor(uniform01(n_rand, seed, st + 2) * H).astype(np.int64)
    target = uniform((n_rand, 3), 0.0, 1.0, seed, st + 3)
    fq = (uniform01(n_rand, seed, st + 4) < 0.5).astype(np.uint8)
    return {"view": views, "px": px, "py": py, "target": target, "fq_mask": fq}


def ray_batch(n_rand: int, seed: int = 0, num_img: int = 30, H: int = H_DEF, W: int = W_DEF,
              focal: float = FOCAL_DEF, step: int = 0):
    """One training batch in the reference's iter_data layout
    (run_lushnerf.py:603-624): rays [N,3,2] (o,d in the last axis), images_idx
    [N,1] int64, target rgb [N,3], fq_mask [N] uint8.  Rays follow the get_rays_np
    formula (helpers:531-539) at pixel centres."""
    c2w = poses(num_img, seed)
    pb = pixel_batch(n_rand, seed, num_img, H, W, step)
    views, px, py = pb["view"], pb["px"].astype(np.float64), pb["py"].astype(np.float64)
    dirs = np.stack([(px + (0.5 - W / 2)) / focal, -(py + (0.5 - H / 2)) / focal,
                     -np.ones_like(px)], -1)
    R = c2w[views][:, :3, :3].astype(np.float64)
    rays_d = np.einsum("nij,nj->ni", R, dirs)
    rays_o = c2w[views][:, :3, 3].astype(np.float64)
    rays = np.stack([rays_o, rays_d], -1).astype(np.float32)          # [N,3,2]
    return {"rays": rays, "images_idx": views.reshape(-1, 1), "target": pb["target"], "fq_mask": pb["fq_mask"]}


def draws(R: int, Ns: int, Ni: int, seed: int = 0, step: int = 0):
    """The four random draws of one march, in the reference's order and shapes
    (models/lushnerf.py:515, :322; helpers:578): t_rand, noise_c, u, noise_f."""
    st = _stream("draws") + 104729 * step
    d = {"t_rand": uniform((R, Ns), 0.0, 1.0, seed, st),
         "noise_c": normal((R, Ns - 1), seed, st + 1)}
    if Ni > 0:
        d["u"] = uniform((R, Ni), 0.0, 1.0, seed, st + 3)
        d["noise_f"] = normal((R, Ns + Ni - 1), seed, st + 4)
    return d
