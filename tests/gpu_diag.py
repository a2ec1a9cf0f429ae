#!/usr/bin/env python3
"""Kernel-by-kernel diagnostics of the HIP path against the CPU oracle (needs a GPU).

Not a pytest file: it runs every check, never stops at the first failure and prints
one line per quantity (normalised max error).  Used while bringing kernels up;
the gating tests are tests/test_gpu_*.py.
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from lush_nerf_amd import lib, ops, synth
from oracle import lush_oracle as O
from tests import util

dev = torch.device("cuda:0")
E2E_PLANES = ops.parse_planes(os.environ.get("LUSH_PLANES", "2,2"))
H, W, F = util.H, util.W, util.FOCAL
RESULTS = []
PER_TENSOR = []      # masked_grad_check appends (tag, tensor, e_gpu, e_f32, 1 - cos gpu, 1 - cos f32) per parameter tensor (tools/parity_table.py)


def rep(name, got, ref, tol):
    e = util.relerr(got, ref)
    ok = e <= tol
    RESULTS.append((name, e, tol, ok))
    print(f"{'ok  ' if ok else 'FAIL'} {name:58s} err={e:.3e} tol={tol:.1e}", flush=True)
    return ok


def section(fn):
    print(f"\n=== {fn.__name__}", flush=True)
    t = time.time()
    try:
        fn()
    except Exception:
        traceback.print_exc()
        RESULTS.append((fn.__name__ + " EXCEPTION", float("inf"), 0, False))
    torch.cuda.synchronize()
    print(f"    ({time.time() - t:.1f}s)", flush=True)


def gpu(x):
    return x.to(dev) if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x)).to(dev)


def batch_of(n, seed):
    b = synth.ray_batch(n, seed, util.NUM_IMG)
    return {k: torch.from_numpy(v) for k, v in b.items()}


def nerf_tensors(p, prefix, D):
    t = []
    for l in range(D):
        t += [p[f"{prefix}.pts_linears.{l}.weight"], p[f"{prefix}.pts_linears.{l}.bias"]]
    for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear"):
        t += [p[f"{prefix}.{n}.weight"], p[f"{prefix}.{n}.bias"]]
    return t


def rbk_tensors(p):
    t = [p["mlp_rbk.view_embedding_layer.view_embed_layer.weight"]]
    for l in range(4):
        t += [p[f"mlp_rbk.view_embed_linears.{l}.weight"], p[f"mlp_rbk.view_embed_linears.{l}.bias"]]
    for n in ("r_branch.0", "v_branch.0", "w_branch.0", "r_linear", "v_linear", "w_linear"):
        t += [p[f"mlp_rbk.{n}.weight"], p[f"mlp_rbk.{n}.bias"]]
    return t


# ------------------------------------------------------------------------------------------
def t_zgrid_pack():
    b = batch_of(96, 3)
    rays = b["rays"].clone().requires_grad_(True)
    ref = O.pack_rays(H, W, F, rays)
    rg = b["rays"].to(dev).requires_grad_(True)
    got = ops.PackRays.apply(rg, H, W, F, True, 0., 1.)
    rep("pack_rays fwd", got, ref, 2e-6)
    g = torch.from_numpy(synth.normal((96, 11), 7))
    ref.backward(g)
    got.backward(g.to(dev))
    rep("pack_rays bwd", rg.grad, rays.grad, 1e-5)
    for S in (32, 64, 128):
        t_rand = torch.from_numpy(synth.uniform((96, S), 0, 1, 5, S))
        zc = O._z_grid(ref[:, 6:7].detach(), ref[:, 7:8].detach(), S, False)
        rep(f"zgrid S={S} det", ops.zgrid(got.detach(), S, False, None), zc, 0.0)
        rep(f"zgrid S={S} jitter", ops.zgrid(got.detach(), S, False, t_rand.to(dev)), O._stratify(zc, t_rand), 1e-7)


def t_gen_rays():
    n = 500
    c2w = synth.poses(30, 2)
    view = (synth.uniform01(n, 3, 1) * 30).astype(np.int64)
    px = np.floor(synth.uniform01(n, 3, 2) * W).astype(np.int64)
    py = np.floor(synth.uniform01(n, 3, 3) * H).astype(np.int64)
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    ro, rd = O.get_rays_np_formula(H, W, F, c2w[view].astype(np.float64), px.astype(np.float64), py.astype(np.float64))
    got = ops.gen_rays(gpu(c2w), gpu(view), gpu(px), gpu(py), K)
    rep("gen_rays origins", got[..., 0], ro, 0.0)
    rep("gen_rays directions", got[..., 1], rd, 2e-7)
    # the bench generator builds its batches with the same formula on the host
    b = synth.ray_batch(64, 9)
    rep("gen_rays vs synth.ray_batch", ops.gen_rays(gpu(synth.poses(30, 9)), gpu(b["images_idx"].reshape(-1)),
        gpu(np.floor(synth.uniform01(64, 9, synth._stream("batch") + 1) * W).astype(np.int64)),
        gpu(np.floor(synth.uniform01(64, 9, synth._stream("batch") + 2) * H).astype(np.int64)), K), b["rays"], 2e-7)


def t_composite():
    # S = 256 is BASELINE config 5's fine pass (128 + 128): composite_fwd_kernel<4> / composite_bwd_kernel<4>, four
    # 64-sample blocks per wavefront with the carry of the exclusive product / the reverse affine scan between them
    for S, train in ((64, True), (128, True), (32, False), (256, True), (256, False), (200, True)):
        R = 70
        raw = torch.from_numpy(synth.normal((R, S, 4), 11, S)) * torch.tensor([1., 1., 1., 30.])
        z = torch.sort(torch.from_numpy(synth.uniform((R, S), 0, 1, 12, S)), -1)[0]
        batch = O.pack_rays(H, W, F, batch_of(R, 4)["rays"])
        noise = torch.from_numpy(synth.normal((R, S - 1), 13, S))
        rawr = raw.clone().requires_grad_(True)
        br = batch.clone().requires_grad_(True)
        ref = O.raw2outputs(rawr, z, br[:, 3:6], 1. if train else 0., False, noise if train else None,
                            training=train, render_rmnearplane=80)
        cfg = ops.MarchCfg(S, 0, 1., 1. if train else 0., near_mask=-1. if train else 80 / 128)
        got = ops.composite_fwd(gpu(raw.reshape(-1, 4)), gpu(z), gpu(batch), gpu(noise) if train else None, cfg)
        tag = f"S={S} {'train' if train else 'eval'}"
        rep(f"composite fwd rgb {tag}", got[0], ref[0], 2e-6)
        rep(f"composite fwd depth {tag}", got[1], ref[4], 2e-6)
        rep(f"composite fwd acc {tag}", got[2], ref[2], 2e-6)
        rep(f"composite fwd weights {tag}", got[3], ref[3], 2e-6)
        rep(f"composite fwd density {tag}", got[4], ref[1], 2e-6)
        g = [torch.from_numpy(synth.normal(s, 14, i)) for i, s in enumerate(((R, 3), (R,), (R,)))]
        (ref[0] * g[0]).sum().add((ref[4] * g[1]).sum()).add((ref[2] * g[2]).sum()).backward()
        drays = torch.zeros(R, 11, device=dev)
        draw = ops.composite_bwd(gpu(raw.reshape(-1, 4)), gpu(z), gpu(batch), gpu(noise) if train else None, cfg,
                                 gpu(g[0]), gpu(g[1]), gpu(g[2]), drays)
        rep(f"composite bwd draw {tag}", draw.view(R, S, 4), rawr.grad, 2e-5)
        rep(f"composite bwd drays_d {tag}", drays[:, 3:6], br.grad[:, 3:6], 2e-5)


def t_sample():
    # (a) the reference's own vectors: bins of this fixture ARE the mid-points of its z (make_golden.case_sample_pdf_z).
    # sample_pdf is discontinuous where a u sits on a CDF knot next to a flat bin (denom < 1e-5 -> 1): with the
    # deterministic u = linspace(0, 1) the last value 1.0 ties with cdf[-1], whose last ulp depends on the summation
    # order (torch.cumsum itself differs between CPU and GPU there).  A sample may therefore differ from the fixture
    # ONLY where the fixture's own CDF has a knot within 4e-7 of that u; everything else must agree to 2e-5.
    gz = util.golden("sample_pdf_z")
    zt, wt, ut = (torch.from_numpy(gz[k]) for k in ("z", "weights", "u"))
    wpdf = wt[:, 1:-1] + 1e-5
    cdf = torch.cumsum(wpdf / wpdf.sum(-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1).double()
    for det in (False, True):
        zo, zs, zstd = ops.sample_merge(gpu(zt), gpu(wt), 64, None if det else gpu(ut))
        tag = "det" if det else "rand"
        uu = (torch.linspace(0., 1., 64).expand(zt.shape[0], 64) if det else ut).double()
        on_knot = (cdf[:, :, None] - uu[:, None, :]).abs().min(1)[0] < 4e-7            # [R, Ni]
        ref_s = torch.from_numpy(gz["s_" + tag])
        err = (zs.cpu() - ref_s).abs() / ref_s.abs().max()
        bad = (err > 2e-5) & ~on_knot
        RESULTS.append((f"sample_pdf reference vectors {tag} (off-knot samples)", float(err[~on_knot].max()), 2e-5, not bool(bad.any())))
        print(f"{'ok  ' if not bad.any() else 'FAIL'} sample_pdf reference vectors {tag}: worst off-knot {float(err[~on_knot].max()):.2e}; "
              f"{int((err > 2e-5).sum())} of {err.numel()} samples differ, all on a CDF knot: {not bool(bad.any())}", flush=True)
        clean = ~(((err > 2e-5) & on_knot).any(-1))                                    # rays without an excused tie
        rep(f"merged sort reference vectors {tag} (rays without a tie)", zo[gpu(clean)], gz["merged_" + tag][clean.numpy()], 2e-5)
        rep(f"z_std reference vectors {tag} (rays without a tie)", zstd[gpu(clean)],
            torch.std(ref_s, -1, unbiased=False)[clean], 2e-5)
    # (b) fresh data against the oracle (which the CPU suite pins to both reference fixtures)
    g = util.golden("sample_pdf")
    R, S = g["bins"].shape[0], g["bins"].shape[1] + 1
    z = torch.sort(torch.from_numpy(synth.uniform((R, S), 0, 1, 21)), -1)[0]
    w = torch.from_numpy(synth.uniform((R, S), 0, 1, 22)) ** 6
    w[:3] = 0
    u = torch.from_numpy(np.minimum(synth.uniform((R, 64), 0, 1, 23), np.float32(1 - 2 ** -24)))
    mid = .5 * (z[:, 1:] + z[:, :-1])
    for det in (False, True):
        ref_s = O.sample_pdf(mid, w[:, 1:-1], 64, det, None if det else u)
        zo, zs, zstd = ops.sample_merge(gpu(z), gpu(w), 64, None if det else gpu(u))
        rep(f"sample_pdf det={det}", zs, ref_s, 2e-5)
        rep(f"merged sort det={det}", zo, torch.sort(torch.cat([z, ref_s], -1), -1)[0], 2e-5)
        rep(f"z_std det={det}", zstd, torch.std(ref_s, -1, unbiased=False), 2e-5)
    srt_ok = bool((zo[:, 1:] >= zo[:, :-1]).all())
    RESULTS.append(("merged output sorted", 0.0 if srt_ok else 1.0, 0, srt_ok))
    print("ok  " if srt_ok else "FAIL", "merged output sorted")
    # (c) every shape class of the sort: whole 64-value rows (sorted in registers up to 256 new samples), more than four
    # rows of new samples (Ni = 320 / 384 / 448: the LDS network -- a lane keeps only four rows), ragged counts.  The merged row
    # must be EXACTLY the sorted union of the depths and of the kernel's own samples (the sort is a permutation: bit for bit);
    # the samples against the oracle on well-conditioned weights (no near-empty bins: section (a) covers those).
    for S2, Ni2 in ((64, 64), (128, 128), (64, 256), (192, 256), (64, 320), (128, 384), (64, 448), (48, 40), (200, 56), (100, 300)):
        z2 = torch.sort(torch.from_numpy(synth.uniform((24, S2), 0, 1, 31 + S2)), -1)[0]
        w2 = torch.from_numpy(synth.uniform((24, S2), 0.2, 1, 32 + Ni2))
        u2 = torch.from_numpy(np.minimum(synth.uniform((24, Ni2), 0, 1, 33 + S2 + Ni2), np.float32(1 - 2 ** -24)))
        ref2 = O.sample_pdf(.5 * (z2[:, 1:] + z2[:, :-1]), w2[:, 1:-1], Ni2, False, u2)
        zo2, zs2, zstd2 = ops.sample_merge(gpu(z2), gpu(w2), Ni2, gpu(u2))
        rep(f"sample_pdf {S2}+{Ni2}", zs2, ref2, 2e-5)
        rep(f"merged sort {S2}+{Ni2} = sorted union of its own samples, bit for bit", zo2, torch.sort(torch.cat([gpu(z2), zs2], -1), -1)[0], 0.0)
        rep(f"z_std {S2}+{Ni2}", zstd2, torch.std(ref2, -1, unbiased=False), 2e-5)


def _mlp_case(net, prefix, D, Wd, seed, R, S, sharp=False):
    p = util.params(seed, sharp=sharp)
    batch = O.pack_rays(H, W, F, batch_of(R, seed)["rays"])
    z = torch.sort(torch.from_numpy(synth.uniform((R, S), 0, 1, seed + 50)), -1)[0]
    return p, batch, z


def t_mlp_fwd():
    for net, prefix, D in ((ops.NET_NERF, "mlp_fine", 8), (ops.NET_NOISE, "mlp_noise_coarse", 4)):
        R, S = (24, 40) if net == ops.NET_NERF else (100, 1)
        p, batch, z = _mlp_case(net, prefix, D, 0, 31, R, S)
        pts = batch[:, None, 0:3] + batch[:, None, 3:6] * z[:, :, None]
        e = O.embed(pts.reshape(-1, 3), 10)
        d = O.embed(batch[:, None, 8:11].expand(R, S, 3).reshape(-1, 3), 4)
        with torch.no_grad():
            full = O.nerf_mlp(p, prefix, torch.cat([e, d], -1), 63, 27, D)
        tens = [gpu(t) for t in nerf_tensors(p, prefix, D)]
        for ns, tol in ((3, 3e-6), (2, 3e-5), (1, 3e-2), (ops.PLANES_F16, 4e-3)):
            pk = ops.mlp_pack(net, ns, tens)
            raw, stash = ops.mlp_forward(net, ns, tens, pk, gpu(batch), gpu(z), True)
            ncmp = 4 if net == ops.NET_NERF else 3
            rep(f"mlp fwd net={net} planes={ns} rgb", raw[:, :3], full[:, :3], tol)
            if net == ops.NET_NERF:
                rep(f"mlp fwd net={net} planes={ns} sigma", raw[:, 3], full[:, 3], tol)
            if ns in (3, 2):   # layer-wise stash check localises a wrong layer (bf16 planes); 3 = tiled kernel, 2 = chain kernel
                stol, ptol = (3e-6, 2e-6) if ns == 3 else (3e-5, 1e-5)   # 2 planes carry ~2^-17 of the value
                off = (lib.C.c_longlong * 16)()
                lib.call("lush_debug_stash_layout", net, ns, R * S, off)
                Ppad, HW = off[12], off[14]
                x = torch.cat([e, d], -1)
                h = x[:, :63]
                sb = stash.cpu().numpy()
                for l in range(D):
                    h = torch.relu(torch.nn.functional.linear(h, p[f"{prefix}.pts_linears.{l}.weight"],
                                                              p[f"{prefix}.pts_linears.{l}.bias"]))
                    arr = np.frombuffer(sb[off[2 + l]:off[2 + l] + ns * Ppad * HW * 2].tobytes(), dtype=np.uint16)
                    planes = (arr.astype(np.uint32) << 16).view(np.float32).reshape(ns, Ppad, HW)
                    rep(f"  stash h{l} net={net} planes={ns}", planes[:, :R * S].sum(0), h.detach().numpy(), stol)
                    if l == 4 and D == 8:
                        h = torch.cat([x[:, :63], h], -1)
                pe = np.frombuffer(sb[off[1]:off[1] + ns * Ppad * 128 * 2].tobytes(), dtype=np.uint16)
                with np.errstate(invalid="ignore"):      # (pad rows and the pad columns 96..127 are never written)
                    pe = (pe.astype(np.uint32) << 16).view(np.float32).reshape(ns, Ppad, 128)[:, :R * S].sum(0)
                rep(f"  stash x (cols 0..2) net={net} planes={ns}", pe[:, :3], e.numpy()[:, :3], 0.0 if ns == 3 else ptol)
                for k in (0, 3, 6, 9):
                    rep(f"  stash sin/cos freq 2^{k} net={net} planes={ns}", pe[:, 3 + 6 * k:9 + 6 * k], e.numpy()[:, 3 + 6 * k:9 + 6 * k], ptol)
                xg = gpu(pts.reshape(-1, 3))
                rep(f"  torch.sin(cuda) vs torch.sin(cpu) @2^9", torch.sin(xg * 512.), torch.sin(pts.reshape(-1, 3) * 512.), 2e-6)
                rep(f"  stash gamma(x) net={net} planes={ns}", pe[:, :63], e.numpy(), ptol)
                rep(f"  stash gamma(d) net={net} planes={ns}", pe[:, 64:91], d.numpy(), ptol)


def t_mlp_ragged():
    """Point counts around the 128-point tile of the chain kernels (1, 127, 129, 300): outputs and input gradients."""
    net, prefix, D = ops.NET_NERF, "mlp_fine", 8
    for R, S in ((1, 1), (127, 1), (43, 3), (3, 100)):
        p, batch, z = _mlp_case(net, prefix, D, 0, 57, R, S)
        pts = batch[:, None, 0:3] + batch[:, None, 3:6] * z[:, :, None]
        x = torch.cat([O.embed(pts.reshape(-1, 3), 10), O.embed(batch[:, None, 8:11].expand(R, S, 3).reshape(-1, 3), 4)], -1)
        with torch.no_grad():
            full = O.nerf_mlp(p, prefix, x, 63, 27, D)
        tens = [gpu(t) for t in nerf_tensors(p, prefix, D)]
        for ns, tol in ((2, 3e-5), (ops.PLANES_F16, 4e-3)):
            pk = ops.mlp_pack(net, ns, tens)
            raw, _ = ops.mlp_forward(net, ns, tens, pk, gpu(batch), gpu(z), True, ops.stash_code(ns, 1))
            rep(f"ragged fwd P={R * S} planes={ns}", raw, full, tol)


def t_mlp_bwd():
    for net, prefix, D in ((ops.NET_NERF, "mlp_coarse", 8), (ops.NET_NOISE, "mlp_noise_coarse", 4)):
        R, S = (20, 48) if net == ops.NET_NERF else (200, 1)
        p, batch, z = _mlp_case(net, prefix, D, 0, 41, R, S)
        p = {k: v.clone().requires_grad_(k.startswith(prefix)) for k, v in p.items()}
        br = batch.clone().requires_grad_(True)
        pts = br[:, None, 0:3] + br[:, None, 3:6] * z[:, :, None]
        x = torch.cat([O.embed(pts.reshape(-1, 3), 10),
                       O.embed(br[:, None, 8:11].expand(R, S, 3).reshape(-1, 3), 4)], -1)
        draw = torch.from_numpy(synth.normal((R * S, 4), 43))
        if net == ops.NET_NOISE:
            draw[:, 3] = 0
        names = [f"pts_linears.{l}.{s}" for l in range(D) for s in ("weight", "bias")] + \
                [f"{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear")
                 for s in ("weight", "bias")]
        tens = [gpu(t.detach()) for t in nerf_tensors(p, prefix, D)]
        H16 = ops.PLANES_F16
        for nf, nb, tol in ((3, 3, 2e-5), (2, 2, 2e-4), (2, 1, 3e-2), (H16, 1, 3e-2), (2, H16, 8e-3), (H16, H16, 2e-3)):   # (2,h): X is the bf16 hi plane (8 bits); (h,h): 11-bit operands throughout
            pk = ops.mlp_pack(net, nf, tens)
            raw, stash = ops.mlp_forward(net, nf, tens, pk, gpu(batch), gpu(z), True, ops.stash_code(nf, nb))
            # reference gradients with the GPU's own ReLU decisions (isolates arithmetic from kink flips)
            masks = util.stash_masks(net, ops.nplanes(ops.stash_code(nf, nb)), R * S, stash, f16=(nf == ops.PLANES_F16))
            for v in list(p.values()) + [br]:
                v.grad = None
            out = util.nerf_mlp_masked(p, prefix, x, D, masks)
            (out * draw).sum().backward(retain_graph=True)
            pkb = pk if nb == nf else ops.mlp_pack(net, nb, tens)
            grads, dpts = ops.mlp_backward(net, ops.stash_code(nf, nb), nb, tens, pkb, gpu(batch), gpu(z), gpu(draw), stash)
            worst, wname = 0.0, ""
            for n, g in zip(names, grads):
                ref = p[f"{prefix}.{n}"].grad
                if ref is None:
                    continue
                e = util.relerr(g, ref)
                if e > worst:
                    worst, wname = e, n
                if nf == 3 or os.environ.get("LUSH_DIAG_ALL"):
                    rep(f"  dW net={net} ({nf},{nb}) {n}", g, ref, tol)
            RESULTS.append((f"mlp bwd net={net} ({nf},{nb}) worst [{wname}]", worst, tol, worst <= tol))
            print(f"{'ok  ' if worst <= tol else 'FAIL'} mlp bwd net={net} planes=({nf},{nb}) worst param grad {wname}: {worst:.3e} tol={tol:.1e}")
            drays = torch.zeros(R, 11, device=dev)
            lib.call("lush_ray_grad_reduce", lib.ptr(dpts), lib.ptr(gpu(z)), R, S, lib.ptr(drays), ops._stream())
            rep(f"mlp bwd net={net} planes=({nf},{nb}) drays", drays, br.grad, tol * 2)


def t_rbk():
    g = util.golden("rbk")
    n, seed = (int(x) for x in g["meta"])
    p = util.params(seed, rbk_scale=3.0e5, requires_grad=True)
    b = batch_of(n, seed)
    rays = b["rays"].clone().requires_grad_(True)
    ref_rays, ref_ccw = O.rbk_forward(p, rays, b["images_idx"])
    tens = [gpu(t.detach()).requires_grad_(True) for t in rbk_tensors(p)]
    rg = gpu(b["rays"]).requires_grad_(True)
    mask = b["fq_mask"]
    got_rays, got_ccw = ops.RbkWarp.apply(rg, gpu(b["images_idx"]), 4, 0.1, gpu(mask), None, *tens)
    rep("rbk new_rays vs golden", got_rays, g["new_rays"], 2e-5)
    rep("rbk ccw vs golden", got_ccw, g["ccw"], 2e-5)
    gr = torch.from_numpy(synth.normal(tuple(ref_rays.shape), 61))
    gc = torch.from_numpy(synth.normal(tuple(ref_ccw.shape), 62))
    m5 = mask.bool().repeat_interleave(5)
    masked = torch.where(m5[:, None, None], ref_rays, ref_rays.detach())
    ((masked * gr).sum() + (ref_ccw * gc).sum()).backward()
    ((got_rays * gpu(gr)).sum() + (got_ccw * gpu(gc)).sum()).backward()
    rep("rbk bwd d rays", rg.grad, rays.grad, 2e-4)
    names = ["embed"] + [f"trunk{l}.{s}" for l in range(4) for s in "wb"] + \
            [f"{n}.{s}" for n in ("r_branch", "v_branch", "w_branch", "r_linear", "v_linear", "w_linear") for s in "wb"]
    for n, t, r in zip(names, tens, rbk_tensors(p)):
        rep(f"  rbk grad {n}", t.grad, r.grad, 3e-4)
    # more images than the LDS-resident tables hold (> 39): the global-memory stages of the same MLP
    n_img, nr = 48, 64
    pw = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.rbk_weights(n_img, 77).items()}
    with torch.no_grad():
        for k in ("mlp_rbk.r_linear.weight", "mlp_rbk.v_linear.weight"):
            pw[k] *= 3.0e5
    bb = synth.ray_batch(nr, 77, n_img)
    rays2 = torch.from_numpy(bb["rays"]).requires_grad_(True)
    idx2 = torch.from_numpy(bb["images_idx"])
    rr, cc = O.rbk_forward(pw, rays2, idx2)
    tens2 = [gpu(t.detach()).requires_grad_(True) for t in rbk_tensors(pw)]
    rg2 = gpu(bb["rays"]).requires_grad_(True)
    gr2, gc2 = ops.RbkWarp.apply(rg2, gpu(idx2), 4, 0.1, None, None, *tens2)
    rep(f"rbk ({n_img} images, global tables) new_rays", gr2, rr, 2e-5)
    rep(f"rbk ({n_img} images, global tables) ccw", gc2, cc, 2e-5)
    g1 = torch.from_numpy(synth.normal(tuple(rr.shape), 63))
    g2 = torch.from_numpy(synth.normal(tuple(cc.shape), 64))
    ((rr * g1).sum() + (cc * g2).sum()).backward()
    ((gr2 * gpu(g1)).sum() + (gc2 * gpu(g2)).sum()).backward()
    worst = max(util.relerr(t.grad, r.grad) for t, r in zip(tens2, rbk_tensors(pw)))
    RESULTS.append((f"rbk ({n_img} images, global tables) worst parameter grad", worst, 3e-4, worst <= 3e-4))
    print(f"{'ok  ' if worst <= 3e-4 else 'FAIL'} rbk ({n_img} images, global tables) worst parameter grad {worst:.2e}")
    # other shapes of the same kernels: fewer motions (narrower heads) and image counts that leave the last workgroup ragged
    for n_img, M in ((5, 2), (1, 1), (7, 3)):
        pw = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.rbk_weights(n_img, 90 + M, num_motion=M).items()}
        with torch.no_grad():
            for k in ("mlp_rbk.r_linear.weight", "mlp_rbk.v_linear.weight"):
                pw[k] *= 3.0e5
        bb = synth.ray_batch(nr, 90 + M, n_img)
        rays3 = torch.from_numpy(bb["rays"]).requires_grad_(True)
        idx3 = torch.from_numpy(bb["images_idx"])
        rr, cc = O.rbk_forward(pw, rays3, idx3, num_motion=M)
        tens3 = [gpu(t.detach()).requires_grad_(True) for t in rbk_tensors(pw)]
        rg3 = gpu(bb["rays"]).requires_grad_(True)
        gr3, gc3 = ops.RbkWarp.apply(rg3, gpu(idx3), M, 0.1, None, None, *tens3)
        rep(f"rbk ({n_img} images, {M} motions) new_rays", gr3, rr, 2e-5)
        rep(f"rbk ({n_img} images, {M} motions) ccw", gc3, cc, 2e-5)
        g1 = torch.from_numpy(synth.normal(tuple(rr.shape), 65))
        g2 = torch.from_numpy(synth.normal(tuple(cc.shape), 66))
        ((rr * g1).sum() + (cc * g2).sum()).backward()
        ((gr3 * gpu(g1)).sum() + (gc3 * gpu(g2)).sum()).backward()
        rep(f"rbk ({n_img} images, {M} motions) d rays", rg3.grad, rays3.grad, 2e-4)
        worst = max(util.relerr(t.grad, r.grad) for t, r in zip(tens3, rbk_tensors(pw)))
        RESULTS.append((f"rbk ({n_img} images, {M} motions) worst parameter grad", worst, 3e-4, worst <= 3e-4))
        print(f"{'ok  ' if worst <= 3e-4 else 'FAIL'} rbk ({n_img} images, {M} motions) worst parameter grad {worst:.2e}")


def t_mix():
    N, M = 50, 5
    x = torch.from_numpy(synth.uniform((N * M, 3), 0.05, 1, 71)).requires_grad_(True)
    ccw = torch.softmax(torch.from_numpy(synth.normal((N, M), 72)), -1).requires_grad_(True)
    nraw = torch.from_numpy(synth.normal((N, 3), 73)).requires_grad_(True)
    tgt = torch.from_numpy(synth.uniform((N, 3), 0, 1, 74))
    pure = O.rbk_weighted_sum(x, ccw)
    a = O.tonemap(pure + 0.1 * torch.sigmoid(nraw))
    b = O.tonemap(pure)
    loss = O.train_loss(a, b, tgt)
    loss.backward()
    xg, cg, ng = (gpu(t.detach()).requires_grad_(True) for t in (x, ccw, nraw))
    pure_g = ops.WSum.apply(xg, cg)
    ag = ops.ToneMap.apply(pure_g, ng, True)
    bg = ops.ToneMap.apply(pure_g, None, True)
    lg = ops.TrainLoss.apply(ag, bg, gpu(tgt))
    lg.backward()
    rep("wsum+tonemap fwd", ag, a, 2e-6)
    rep("loss", lg.reshape(1), loss.reshape(1), 2e-6)
    rep("mix bwd dx", xg.grad, x.grad, 2e-5)
    rep("mix bwd dccw", cg.grad, ccw.grad, 2e-5)
    rep("mix bwd dnoise", ng.grad, nraw.grad, 2e-5)
    y = ops.NoiseAct.apply(ng.detach().requires_grad_(True))
    rep("noise_act", y, 0.1 * torch.sigmoid(nraw), 2e-6)


def t_warp_ndc():
    """ops.RbkWarpNdc (rbk_warp_ndc_fwd / _bwd_kernel: warp + NDC + view direction + pack in one kernel per direction) against
    the oracle's rbk_forward followed by pack_rays (models/lushnerf.py:75-98, 118-153, 772-795; helpers:542-562), forward and
    every gradient, with the allkernel mask; and against the piecewise ops it replaces (RbkWarp -> PackRays)."""
    g = util.golden("rbk")
    n, seed = (int(x) for x in g["meta"])
    b = batch_of(n, seed)
    mask = b["fq_mask"]
    for ndc in (True, False):
        p = util.params(seed, rbk_scale=3.0e5, requires_grad=True)
        rays = b["rays"].clone().requires_grad_(True)
        ref_rays, ref_ccw = O.rbk_forward(p, rays, b["images_idx"])
        m5 = mask.bool().repeat_interleave(5)
        masked = torch.where(m5[:, None, None], ref_rays, ref_rays.detach())
        ref_batch = O.pack_rays(H, W, F, masked) if ndc else O.pack_rays(H, W, F, masked, ndc=False, near=2., far=6.)
        near, far = (0., 1.) if ndc else (2., 6.)
        tens = [gpu(t.detach()).requires_grad_(True) for t in rbk_tensors(p)]
        rg = gpu(b["rays"]).requires_grad_(True)
        batch, ccw, batch0 = ops.RbkWarpNdc.apply(rg, gpu(b["images_idx"]), 4, 0.1, gpu(mask), None, H, W, F, ndc, near, far, *tens)
        tag = "ndc" if ndc else "no ndc"
        rep(f"warp_ndc batch ({tag})", batch, ref_batch, 2e-5)
        rep(f"warp_ndc ccw ({tag})", ccw, ref_ccw, 2e-5)
        rep(f"warp_ndc batch0 = rows of the input rays ({tag})", batch0, batch.detach().view(n, 5, 11)[:, 0], 0.0)
        gb = torch.from_numpy(synth.normal(tuple(ref_batch.shape), 65)) * torch.tensor([1., 1., 1., 1., 1., 1., 0., 0., 1., 1., 1.])
        gc = torch.from_numpy(synth.normal(tuple(ref_ccw.shape), 66))
        ((ref_batch * gb).sum() + (ref_ccw * gc).sum()).backward()
        ((batch * gpu(gb)).sum() + (ccw * gpu(gc)).sum()).backward()
        rep(f"warp_ndc bwd d rays ({tag})", rg.grad, rays.grad, 2e-4)
        worst = max(util.relerr(t.grad, r.grad) for t, r in zip(tens, rbk_tensors(p)))
        RESULTS.append((f"warp_ndc worst parameter grad ({tag})", worst, 3e-4, worst <= 3e-4))
        print(f"{'ok  ' if worst <= 3e-4 else 'FAIL'} warp_ndc worst parameter grad ({tag}) {worst:.2e}")
        # the piecewise ops on the same inputs: same numbers (the fused kernel runs the same device functions)
        tens2 = [gpu(t.detach()).requires_grad_(True) for t in rbk_tensors(p)]
        rg2 = gpu(b["rays"]).requires_grad_(True)
        nr, cc = ops.RbkWarp.apply(rg2, gpu(b["images_idx"]), 4, 0.1, gpu(mask), None, *tens2)
        pb = ops.PackRays.apply(nr, H, W, F, ndc, near, far)
        rep(f"warp_ndc vs RbkWarp -> PackRays, batch ({tag})", batch, pb, 1e-6)
        ((pb * gpu(gb)).sum() + (cc * gpu(gc)).sum()).backward()
        rep(f"warp_ndc vs RbkWarp -> PackRays, d rays ({tag})", rg.grad, rg2.grad, 2e-5)
        worst = max(util.relerr(t.grad, r.grad) for t, r in zip(tens, tens2))
        RESULTS.append((f"warp_ndc vs piecewise worst parameter grad ({tag})", worst, 2e-5, worst <= 2e-5))
        print(f"{'ok  ' if worst <= 2e-5 else 'FAIL'} warp_ndc vs piecewise worst parameter grad ({tag}) {worst:.2e}")


def t_blur_mix():
    """ops.BlurMix (blur_mix_fwd / _bwd_kernel) against the oracle's rbk_weighted_sum + 0.1 sigmoid + tonemap
    (models/lushnerf.py:644-654, 100-116; helpers:164-174): the five outputs, and the gradients of a loss that uses all five."""
    N, M = 50, 5
    for gamma in (True, False):
        x = torch.from_numpy(synth.uniform((N * M, 3), 0.05, 1, 71)).requires_grad_(True)
        x0 = torch.from_numpy(synth.uniform((N * M, 3), 0.05, 1, 75)).requires_grad_(True)
        ccw = torch.softmax(torch.from_numpy(synth.normal((N, M), 72)), -1).requires_grad_(True)
        nraw = torch.from_numpy(synth.normal((N, 3), 73)).requires_grad_(True)
        tm = O.tonemap if gamma else (lambda v: v)
        pure, pure0 = O.rbk_weighted_sum(x, ccw), O.rbk_weighted_sum(x0, ccw)
        nz = 0.1 * torch.sigmoid(nraw)
        ref = [tm(pure + nz), tm(pure0 + nz), nz, tm(pure), tm(pure0)]
        gs = [torch.from_numpy(synth.normal((N, 3), 80 + i)) for i in range(5)]
        sum((r * g).sum() for r, g in zip(ref, gs)).backward()
        xg, x0g, cg, ng = (gpu(t.detach()).requires_grad_(True) for t in (x, x0, ccw, nraw))
        got = ops.BlurMix.apply(xg, x0g, cg, ng, gamma)
        tag = "gamma" if gamma else "none"
        for name, a, r in zip(("blur", "blur0", "noise", "sharp", "sharp0"), got, ref):
            rep(f"blur_mix {name} ({tag})", a, r, 2e-6)
        sum((a * gpu(g)).sum() for a, g in zip(got, gs)).backward()
        rep(f"blur_mix d rgb ({tag})", xg.grad, x.grad, 2e-5)
        rep(f"blur_mix d rgb0 ({tag})", x0g.grad, x0.grad, 2e-5)
        rep(f"blur_mix d ccw ({tag})", cg.grad, ccw.grad, 2e-5)
        rep(f"blur_mix d noise_raw ({tag})", ng.grad, nraw.grad, 2e-5)
    # the trainer's case: gradients for blur and blur0 only (the others arrive as None)
    xg, x0g, cg, ng = (gpu(t.detach()).requires_grad_(True) for t in (x, x0, ccw, nraw))
    got = ops.BlurMix.apply(xg, x0g, cg, ng, True)
    tgt = torch.from_numpy(synth.uniform((N, 3), 0, 1, 74))
    x_, x0_, c_, n_ = (t.detach().clone().requires_grad_(True) for t in (x, x0, ccw, nraw))
    nz = 0.1 * torch.sigmoid(n_)
    O.train_loss(O.tonemap(O.rbk_weighted_sum(x_, c_) + nz), O.tonemap(O.rbk_weighted_sum(x0_, c_) + nz), tgt).backward()
    ops.TrainLoss.apply(got[0], got[1], gpu(tgt)).backward()
    rep("blur_mix + loss d rgb", xg.grad, x_.grad, 2e-5)
    rep("blur_mix + loss d ccw", cg.grad, c_.grad, 2e-5)
    rep("blur_mix + loss d noise_raw", ng.grad, n_.grad, 2e-5)


def t_march_e2e():
    from lush_nerf_amd import model as M
    import argparse
    for name in ("rays_c1_train", "rays_6464_train_sharp", "rays_6464_eval_sharp"):
        g = util.golden(name)
        n, Ns, Ni, train, sharp, seed = (int(x) for x in g["meta"])
        args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                                  N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                                  rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                                  render_rmnearplane=80)
        rbk = M.RBK(util.NUM_IMG, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4)
        net = M.NeRFAll(args, rbk, precision=ops.Precision(*E2E_PLANES))
        w = synth.all_weights(util.NUM_IMG, seed, sharp=bool(sharp))
        if Ni == 0:
            w = {k: v for k, v in w.items() if not k.startswith("mlp_fine.")}
        M.load_reference_weights(net, w)
        net = net.to(dev).train(bool(train))
        b = batch_of(n, seed)
        K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
        draws = {k: v.to(dev) for k, v in util.tdraws(n, Ns, Ni, seed).items()} if train else None
        with torch.no_grad():
            (rgb, depth, acc, extras), noise = net.render_infer(
                H, W, K, 1 << 20, rays=gpu(b["rays"]), perturb=1. if train else 0., N_importance=Ni, N_samples=Ns,
                use_viewdirs=True, white_bkgd=False, raw_noise_std=1. if train else 0., inference=not train,
                near=0., far=1., retraw=True, draws=draws)
        rep(f"e2e {name} rgb_map", rgb, g["rgb_map"], 1e-4)
        rep(f"e2e {name} depth_map", depth, g["depth_map"], 1e-3)
        rep(f"e2e {name} acc_map", acc, g["acc_map"], 1e-4)
        rep(f"e2e {name} noise_rgb", noise, g["noise_rgb"], 1e-4)
        if Ni > 0:
            rep(f"e2e {name} rgb0", extras["rgb0"], g["rgb0"], 1e-4)
            # sample_pdf is discontinuous in its inputs: where a (deterministic, eval) u falls on a CDF step, a 1-ulp
            # difference of the coarse weights moves one of the 64 fine samples by a whole bin and z_std of THAT ray
            # by ~1e-2 (seen: 1 of the 48 eval rays after the alpha head's accumulation order changed, rgb within 3e-7).
            # Gate: all rays within 5e-2, and at most max(1, 0.5 %) of the rays beyond 2e-3.
            zs, zr = extras["z_std"].detach().cpu().numpy().astype(np.float64), np.asarray(g["z_std"], dtype=np.float64)
            dz = np.abs(zs - zr) / max(np.abs(zr).max(), 1e-30)
            rep(f"e2e {name} z_std (worst ray)", extras["z_std"], g["z_std"], 5e-2)
            nbad, allowed = int((dz > 2e-3).sum()), max(1, int(5e-3 * dz.size))
            ok = nbad <= allowed
            if nbad and not train:
                # the claim behind this gate, checked: every ray that moved has a deterministic u (a linspace value)
                # sitting on a step of the coarse CDF within a few ulp, where searchsorted's answer is ill-conditioned
                p_or = util.params(seed, sharp=bool(sharp))
                with torch.no_grad():
                    bo = O.pack_rays(H, W, F, b["rays"])
                    ro = O.render_rays(p_or, bo, Ns, perturb=0., N_importance=0, raw_noise_std=0., training=False,
                                       render_rmnearplane=80, with_noise_branch=False)
                wts = ro["_weights"][:, 1:-1] + 1e-5
                cdf = torch.cumsum(wts / wts.sum(-1, keepdim=True), -1)
                cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1).double()
                u = torch.linspace(0., 1., Ni).double()
                for r in np.nonzero(dz > 2e-3)[0]:
                    gap = float((cdf[r][:, None] - u[None, :]).abs().min())
                    on_step = gap < 4e-7
                    RESULTS.append((f"e2e {name} ray {int(r)}: moved sample sits on a CDF step", gap, 4e-7, on_step))
                    print(f"{'ok  ' if on_step else 'FAIL'} e2e {name} ray {int(r)} z_std moved {dz[r]:.2e}: min |cdf_k - u_i| = {gap:.2e}", flush=True)
            RESULTS.append((f"e2e {name} z_std (rays beyond 2e-3)", float(nbad), float(allowed), ok))
            print(f"{'ok  ' if ok else 'FAIL'} {'e2e ' + name + ' z_std (rays beyond 2e-3, of ' + str(dz.size) + ')':58s} n={nbad} allowed={allowed}", flush=True)


def _nerf_all(Ni=64, seed=0, sharp=True, precision=None, rbk_scale=1.0, train=True, trained_like=False):
    from lush_nerf_amd import model as M
    import argparse
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)
    net = M.NeRFAll(args, M.RBK(util.NUM_IMG, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                    precision=precision or ops.Precision(*E2E_PLANES))
    M.load_reference_weights(net, synth.all_weights(util.NUM_IMG, seed, sharp=sharp, rbk_scale=rbk_scale, trained_like=trained_like))
    return net.to(dev).train(train)


def nondc_batch(n, seed):
    b = synth.ray_batch(n, seed, util.NUM_IMG)
    o, d = b["rays"][..., 0], b["rays"][..., 1]
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    ones = np.ones((n, 1), np.float32)
    return torch.from_numpy(np.concatenate([o, d, 2.0 * ones, 6.0 * ones, vd], -1).astype(np.float32))


def t_lindisp_white():
    """lindisp=True + white_bkgd=True (models/lushnerf.py:393-396, 349-350) against the reference fixture, forward and
    (masked oracle) backward."""
    g = util.golden("rays_lindisp_white")
    n, Ns, Ni, seed = (int(x) for x in g["meta"])
    net = _nerf_all(Ni, seed)
    batch = nondc_batch(n, seed)
    cpu_draws = util.tdraws(n, Ns, Ni, seed)
    bg = gpu(batch).requires_grad_(True)
    keep = net.hooks.keep = {}
    try:
        ret, ret_noise = net.render_rays(bg, N_samples=Ns, retraw=True, lindisp=True, perturb=1., N_importance=Ni,
                                         white_bkgd=True, raw_noise_std=1., draws={k: v.to(dev) for k, v in cpu_draws.items()})
    finally:
        net.hooks.keep = None
    for k, tol in (("rgb_map", 1e-4), ("rgb0", 1e-4), ("acc_map", 1e-4), ("acc0", 1e-4), ("depth_map", 1e-3), ("depth0", 1e-3),
                   ("z_std", 2e-3)):
        rep(f"lindisp+white {k}", ret[k], g[k], tol)
    rep("lindisp+white noise rgb", ret_noise["rgb_map"], g["noise_rgb"], 1e-4)
    G = [gpu(synth.normal((n, 3), 91, i)) for i in range(2)]
    ((ret["rgb_map"] * G[0]).sum() + (ret["rgb0"] * G[1]).sum() + ret["depth_map"].sum() + ret["acc_map"].sum()).backward()
    def run_oracle(dt):
        p = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=True).items()}
        bc = batch.to(dt).clone().requires_grad_(True)
        ro, _ = O.render_rays(p, bc, Ns, retraw=True, lindisp=True, perturb=1., N_importance=Ni, white_bkgd=True,
                              raw_noise_std=1., draws={k: v.to(dt) for k, v in cpu_draws.items()})
        ((ro["rgb_map"] * G[0].cpu().to(dt)).sum() + (ro["rgb0"] * G[1].cpu().to(dt)).sum() + ro["depth_map"].sum() + ro["acc_map"].sum()).backward()
        return p, {"d ray batch": bc}
    gg = {k: v for k, v in _canon_grads(net).items() if not k.startswith("mlp_rbk.") and not k.startswith("mlp_noise")}
    # columns 6, 7 (near, far) carry no gradient on the GPU path (z depends on them only through detached bounds in
    # the trainer's use); compare o, d, viewdir
    cols = [0, 1, 2, 3, 4, 5, 8, 9, 10]

    class _Cols:      # view of the oracle leaf restricted to the compared columns
        def __init__(self, t):
            self.grad = t.grad[:, cols]

    def run_oracle_cols(dt):
        p, x = run_oracle(dt)
        return p, {"d ray batch": _Cols(x["d ray batch"])}
    masked_grad_check("lindisp+white", run_oracle_cols, gg, {"d ray batch": bg.grad[:, cols]}, keep, net.precision)


def _canon_grads(net):
    grads = {}
    for k, v in net.named_parameters():
        ck = k
        if k.startswith("blur_kernel_net.RBK."):
            ck = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        elif k.startswith("blur_kernel_net.view_embed_layer."):
            ck = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        grads[ck] = v.grad
    return grads


def t_eval_forward():
    """NeRFAll.forward(poses=...) in eval mode against the reference fixture (render_path over every
    pixel through lush_gen_rays_image, near-plane mask, 0.1*sigmoid noise image, tone map)."""
    g = util.golden("eval_forward")
    Hh, Ww, seed = (int(x) for x in g["meta"])
    Ff = float(g["focal"])
    K = [[Ff, 0, Ww / 2], [0, Ff, Hh / 2], [0, 0, 1]]
    net = _nerf_all(64, seed, train=False)
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, near=0., far=1.)
    rgbs, noise, depths = net(Hh, Ww, K, chunk=128, poses=gpu(synth.poses(2, seed)), render_kwargs=rk)
    rep("eval forward rgbs", rgbs, g["rgbs"], 1e-4)
    rep("eval forward noise image", noise, g["noise"], 1e-4)
    rep("eval forward depths", depths, g["depths"], 1e-3)
    # device-side image rays against the host formula of the reference
    c2w = torch.from_numpy(synth.poses(2, seed))[1]
    ro, rd = O.get_rays(Hh, Ww, K, c2w)
    img = ops.gen_rays_image(gpu(c2w), Hh, Ww, K)
    rep("gen_rays_image origins", img[..., 0], ro, 0.0)
    rep("gen_rays_image directions", img[..., 1], rd, 2e-7)


def consistency_inputs(g):
    V, ns, seed, anchor = (int(x) for x in g["meta"])
    HW = H * W
    st = torch.from_numpy(g["samples"])
    am = torch.zeros(V, HW, 4)
    am[:, st, 2] = torch.from_numpy(g["ax"])
    am[:, st, 3] = torch.from_numpy(g["ay"])
    cm = torch.zeros(V, HW, dtype=torch.bool)
    cm[:, st] = torch.from_numpy(g["cert_in"]) != 0
    return V, ns, seed, anchor, st, am, cm


def t_consistency():
    """NeRFAll.forward(consist_loss=True) -> Render_Aligned_Pixel (lush_align_rays + one march) and the
    masked-L1 consistency term (lush_consist_loss_fwd_bwd), against the reference fixture and the masked oracle."""
    g = util.golden("consistency")
    V, ns, seed, anchor, st, am, cm = consistency_inputs(g)
    net = _nerf_all(64, seed)
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, save_warped_ray_img=False, near=0., far=1.)
    poses = torch.from_numpy(synth.poses(V, seed))
    keep = net.hooks.keep = {}
    try:     # the tables are indexed by the anchor only: {anchor: [V, HW, ...]} stands in for the [V, V, HW, ...] tensors
        rgb_align, cert = net(H, W, K, 1 << 20, poses=gpu(poses), render_kwargs=rk, render_factor=0,
                              rays_info=torch.arange(V), consist_loss=True, Align_matrix={anchor: gpu(am)},
                              Align_mask={anchor: gpu(cm)}, anchor_pose=anchor, samples=st)
    finally:
        net.hooks.keep = None
    rep("consistency rgb_align", rgb_align, g["rgb_align"], 1e-4)
    rep("consistency align_certainty", cert, g["certainty"], 0.0)
    loss = ops.ConsistLoss.apply(rgb_align, cert, 0.8)
    loss.backward()
    rep("consistency loss_rgb", loss.reshape(1), g["loss_rgb"].reshape(1), 1e-5)
    # float certainty table takes the same path
    rg2, cert2 = ops.align_rays(gpu(poses), gpu(am), gpu(cm.float() * 0.9), st, H, W, K)
    rep("consistency certainty from a float table", cert2, g["certainty"] * 0.9, 1e-7)
    grads = _canon_grads(net)
    none_ref = set(str(x) for x in g["grad_none"])
    none_got = set(k for k, v in grads.items() if v is None)
    ok = none_ref == none_got
    RESULTS.append(("consistency grad-None set", 0. if ok else 1., 0, ok))
    print("ok  " if ok else "FAIL", "consistency grad None set", sorted(none_ref ^ none_got)[:6])
    def run_oracle(dt):
        p = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=True).items()}
        ra, ca = O.render_aligned_pixel(p, H, W, F, poses.to(dt), am.to(dt), cm, st, 64, 64)
        O.consist_loss(ra, ca.to(dt), 0.8).backward()
        return p, {}
    masked_grad_check("consistency", run_oracle, grads, {}, keep, net.precision)
    w2 = util.check_grads({k: v for k, v in grads.items() if v is not None}, g, 1e9)
    wk = max(w2, key=w2.get)
    sec = 3e-2 if E2E_PLANES[0] in (2, 3) else 1e-1
    RESULTS.append((f"consistency worst grad vs fixture [{wk}]", w2[wk], sec, bool(w2[wk] <= sec)))
    print(f"{'ok  ' if w2[wk] <= sec else 'FAIL'} consistency worst grad vs reference fixture {wk}: {w2[wk]:.3e}")
    # the loss kernel alone, against torch autograd through compute_mean_with_confidence
    rr = torch.from_numpy(synth.uniform((7, 32, 3), 0, 1, 77)).requires_grad_(True)
    cc = torch.from_numpy(synth.uniform((7, 32), 0.5, 1, 78))
    cc[:, 3] = 0.1
    lo = O.consist_loss(rr, cc, 0.8)
    lo.backward()
    rg = gpu(rr.detach()).requires_grad_(True)
    lg = ops.ConsistLoss.apply(rg, gpu(cc), 0.8)
    lg.backward()
    rep("consist loss kernel fwd", lg.reshape(1), lo.reshape(1), 2e-6)
    rep("consist loss kernel bwd", rg.grad, rr.grad, 2e-6)


def t_draws():
    """lush_draws: the four draws of a march from one Philox launch -- ranges, moments, independence, determinism."""
    R, Ns, Ni = 4096, 64, 64
    hk = ops.Hooks()
    d1 = ops.march_draws(R, Ns, Ni, 1., 1., dev, seed=1234, hooks=hk)
    d1b = ops.march_draws(R, Ns, Ni, 1., 1., dev, seed=1234, offset=hk.draw_offset)
    d2 = ops.march_draws(R, Ns, Ni, 1., 1., dev, seed=1234, hooks=hk)
    shapes_ok = d1["t_rand"].shape == (R, Ns) and d1["noise_c"].shape == (R, Ns - 1) and d1["u"].shape == (R, Ni) and \
        d1["noise_f"].shape == (R, Ns + Ni - 1)
    RESULTS.append(("draws: reference shapes", 0. if shapes_ok else 1., 0, shapes_ok))
    det = all(torch.equal(d1[k], d1b[k]) for k in d1)
    RESULTS.append(("draws: same (seed, offset) -> same values", 0. if det else 1., 0, det))
    fresh = all(not torch.equal(d1[k], d2[k]) for k in d1)
    RESULTS.append(("draws: next offset -> new values", 0. if fresh else 1., 0, fresh))
    for k in ("t_rand", "u"):
        x = d1[k].double()
        ok = float(x.min()) >= 0. and float(x.max()) < 1.
        RESULTS.append((f"draws {k} in [0, 1)", 0. if ok else 1., 0, ok))
        n = x.numel()
        RESULTS.append((f"draws {k} mean", abs(float(x.mean()) - 0.5), 4 * (1 / 12 / n) ** 0.5, abs(float(x.mean()) - 0.5) < 4 * (1 / 12 / n) ** 0.5))
        RESULTS.append((f"draws {k} variance", abs(float(x.var()) - 1 / 12), 2e-3, abs(float(x.var()) - 1 / 12) < 2e-3))
        # Kolmogorov distance to U[0,1)
        xs = torch.sort(x.reshape(-1))[0]
        ks = float((xs - torch.arange(1, n + 1, device=dev, dtype=torch.float64) / n).abs().max())
        RESULTS.append((f"draws {k} Kolmogorov distance", ks, 2.0 / n ** 0.5, ks < 2.0 / n ** 0.5))
    for k in ("noise_c", "noise_f"):
        x = d1[k].double()
        n = x.numel()
        m, v = float(x.mean()), float(x.var())
        kurt = float(((x - m) ** 4).mean() / v ** 2)
        RESULTS.append((f"draws {k} mean", abs(m), 4 / n ** 0.5, abs(m) < 4 / n ** 0.5))
        RESULTS.append((f"draws {k} variance", abs(v - 1), 1e-2, abs(v - 1) < 1e-2))
        RESULTS.append((f"draws {k} kurtosis", abs(kurt - 3), 5e-2, abs(kurt - 3) < 5e-2))
        xs = torch.sort(x.reshape(-1))[0]
        cdf = 0.5 * (1 + torch.erf(xs / 2 ** 0.5))
        ks = float((cdf - torch.arange(1, n + 1, device=dev, dtype=torch.float64) / n).abs().max())
        RESULTS.append((f"draws {k} Kolmogorov distance to N(0,1)", ks, 2.0 / n ** 0.5, ks < 2.0 / n ** 0.5))
    c = float(torch.corrcoef(torch.stack([d1["t_rand"].reshape(-1)[:200000], d1["u"].reshape(-1)[:200000]]))[0, 1])
    RESULTS.append(("draws: arrays uncorrelated", abs(c), 1e-2, abs(c) < 1e-2))
    for r in RESULTS[-19:]:
        print(f"{'ok  ' if r[3] else 'FAIL'} {r[0]:58s} {r[1]:.3e} (<= {r[2]:.1e})", flush=True)


def t_faults():
    """The numerical-fault word (models/lushnerf.py:474-478, 578-582 print per key; here one device word)."""
    from lush_nerf_amd import lib as L
    n, Ns, Ni, seed = 16, 64, 64, 5
    b = batch_of(n, seed)
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    kw = dict(perturb=1., N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1.,
              inference=False, near=0., far=1., retraw=True)

    def word(mutate):
        net = _nerf_all(Ni, seed)
        with torch.no_grad():
            mutate(net)
            net.render_infer(H, W, K, 1 << 20, rays=gpu(b["rays"]), **kw)
        return net.read_faults()
    clean = word(lambda net: None)
    RESULTS.append(("fault word of a clean run", float(clean), 0, clean == 0))
    w1 = word(lambda net: net.mlp_fine.rgb_linear.bias.fill_(float("nan")))
    ok1 = (w1 & (L.FAULT_BITS["rgb_map"] | L.FAULT_BITS["raw"])) == (L.FAULT_BITS["rgb_map"] | L.FAULT_BITS["raw"]) and not (w1 & L.FAULT_BITS["rgb0"])
    RESULTS.append((f"NaN in the fine rgb head -> rgb_map + raw bits ({L.fault_names(w1)})", float(w1), 0, ok1))
    w2 = word(lambda net: net.mlp_coarse.alpha_linear.bias.fill_(float("inf")))
    ok2 = bool(w2 & L.FAULT_BITS["density0"]) and bool(w2 & L.FAULT_BITS["raw0"])
    RESULTS.append((f"Inf in the coarse sigma head -> density0 + raw0 bits ({L.fault_names(w2)})", float(w2), 0, ok2))
    for r in RESULTS[-3:]:
        print("ok  " if r[3] else "FAIL", r[0], flush=True)


# PRIMARY end-to-end gradient gate.  The oracle is evaluated with every MLP ReLU replaced by the decision the GPU took
# (read back from its stash), once in fp32 and once in float64.  The float64 run is the yardstick: per parameter tensor
#     e_gpu = |grad_gpu - grad_f64|_max / |grad_f64|_max        e_f32 = the same for the fp32 oracle (= the reference's
#     own arithmetic; large for tensors whose gradient is a cancelling sum, e.g. the 1-element alpha bias)
# and a tensor passes when  e_gpu <= max(FLOOR[mode], FACTOR * e_f32).  FLOOR is the mode's arithmetic floor on
# well-conditioned tensors: (2,2)/(3,3) split-bf16 products are fp32-equivalent -> 2e-4; one bf16 plane in the backward
# (x,1) -> 4e-2, the measured bound of 8-bit operands through ten layers (2.6e-2 worst case seen).  FACTOR covers the
# tensors on which fp32 itself is ill-conditioned (gradients that are cancelling sums): the same condition number
# amplifies the forward's operand rounding, 2^-17 with two bf16 planes against fp32's 2^-24, so such a tensor may sit at
# up to 32 x the fp32 oracle's own error (measured: 12 x on the alpha bias of the sharp fixtures).
MASKED_FLOOR = {(3, 3): 1e-4, (2, 2): 2e-4, (2, 1): 4e-2, (ops.PLANES_F16, 1): 4e-2, (1, 1): 8e-2,
                (2, ops.PLANES_F16): 1e-2, (ops.PLANES_F16, ops.PLANES_F16): 5e-3}
MASKED_FACTOR = 32.0
# one fp16 plane in the forward carries 2^-11 per operand: the same cancelling sums sit at up to ~2^13 x fp32's error
# (measured 370 x .. 1100 x on the 1-element alpha bias, 2.6e-2 .. 7.7e-2 absolute, identical with the bf16 and the fp16
# backward: it is the forward's rounding that the condition number amplifies).  Capped in absolute terms.  For these
# modes the training-trajectory test (tests/test_gpu_parity.py) is the arbiter of whether such gradients train alike.
MASKED_FACTOR_F16_FWD = 2048.0
MASKED_CAP = 1e-1
# Round 5: the cap per case = 2 x the worst tensor seen there in (h,h) / (h,1) (profiles/r05_parity_summary.md), never above the
# round figure: the bench's own regime and the two steps added this round are held to what was measured, not to 1e-1.  The
# fixtures whose worst tensor is the 1-element alpha bias of a sharp net (lindisp + white: 8.1e-2) keep 1e-1 = 1.23 x the worst seen.
MASKED_CAP_BY_TAG = {"bench regime": 2.5e-2, "bench regime 128+128": 1e-2, "c1 step": 5e-2, "consist step": 3.5e-2,
                     "consistency": 1e-2, "train train_naive_sharp": 3e-2, "train train_kernel_sharp": 3e-2,
                     "train train_kernel_default": 6.5e-2}
MASKED_GATE = MASKED_FLOOR      # (name kept for the sections that only need the floor)
# Round 6: the step from the trained-like density field (synth.all_weights(trained_like=True)).  Behind a x3000 density head the
# derivative with respect to the RAY GEOMETRY is a cancelling sum over sharp surfaces: the fp32 oracle itself is 0.6 .. 1.5 % from
# float64 on d(rays) and on the blur-kernel network's tensors (the sharp reference fixtures: 1e-4 .. 1e-3), and one fp16 plane in
# the forward sits at 13 .. 21 % there (16 x fp32's miss; 1 - cos 1.2e-2), two bf16 planes at 1.7 % (1 - cos 5.8e-5); the 8 x 256
# networks' own tensors: median 9e-3 / 7e-5.  Gates of that regime = 2 .. 3 x what was measured (profiles/r06_trained_like_grads.md).
MASKED_CAP_BY_TAG["trained-like regime"] = 5e-1
COS_GATE_BY_TAG = {"trained-like regime": {(ops.PLANES_F16, ops.PLANES_F16): 3e-2, (2, ops.PLANES_F16): 3e-3, (2, 2): 2e-4}}
WELL_FLOOR_BY_TAG = {"trained-like regime": {(ops.PLANES_F16, ops.PLANES_F16): 1.2e-2}}


def masked_grad_check(tag, run_oracle, gpu_grads, gpu_extra, keep, prec):
    """run_oracle(dtype) -> (params dict with .grad, {name: extra leaf tensors with .grad}) evaluated INSIDE a
    masked_oracle context; gpu_grads {canonical name: tensor or None}; gpu_extra {name: tensor} (e.g. d rays)."""
    masks = _gpu_masks(keep, prec)
    res = {}
    flips = None
    for dt in (torch.float64, torch.float32):
        with util.masked_oracle(masks) as mo:
            res[dt] = run_oracle(dt)
        if dt == torch.float32:
            flips = mo.flips
    (p64, x64), (p32, x32) = res[torch.float64], res[torch.float32]
    floor = MASKED_FLOOR.get(tuple(E2E_PLANES), 3e-2)
    factor = MASKED_FACTOR_F16_FWD if E2E_PLANES[0] == ops.PLANES_F16 else MASKED_FACTOR
    worst_excess, worst = 0.0, ("", 0.0, 0.0)
    items = [(k, v, p64[k].grad, p32[k].grad) for k, v in gpu_grads.items() if v is not None]
    items += [(k, v, x64[k].grad, x32[k].grad) for k, v in gpu_extra.items() if v is not None and x64[k].grad is not None]
    for k, got, t64, t32 in items:
        e_gpu, e_f32 = util.relerr(got, t64), util.relerr(t32, t64)
        PER_TENSOR.append((tag, k, e_gpu, e_f32))
        excess = e_gpu / max(floor, min(MASKED_CAP_BY_TAG.get(tag, MASKED_CAP), factor * e_f32))
        if not excess <= worst_excess:          # NaN propagates
            worst_excess, worst = excess, (k, e_gpu, e_f32)
    ok = bool(worst_excess <= 1.0)
    RESULTS.append((f"{tag} MASKED grads vs float64 [{worst[0]}] (e_gpu / allowed)", worst_excess, 1.0, ok))
    print(f"{'ok  ' if ok else 'FAIL'} {tag} masked oracle: worst tensor {worst[0]}: e_gpu {worst[1]:.2e}, fp32 oracle {worst[2]:.2e}, "
          f"allowed max({floor:.0e}, min({MASKED_CAP_BY_TAG.get(tag, MASKED_CAP):.1e}, {factor:.0f} x fp32)) -> {worst_excess:.2f} of the allowance", flush=True)
    # direction of every gradient tensor: cosine with the float64 oracle's.  A gate that does not scale with the mode's
    # element-wise allowance: whatever the operand width, a parameter tensor must be pushed the way float64 pushes it --
    # 0.999 with 16-bit operands in the backward, 0.999999 fp32-equivalent; tensors on which the fp32 oracle itself
    # is off by more than (1 - cos) / 4 (cancelling sums such as a 1-element bias) get that much slack.
    def cosine(a, b):
        a, b = torch.as_tensor(a).detach().double().cpu().reshape(-1), torch.as_tensor(b).detach().double().cpu().reshape(-1)
        na, nb = float(a.norm()), float(b.norm())
        return 1.0 if na == 0.0 and nb == 0.0 else float(a @ b) / max(na * nb, 1e-300)
    cos_gate = COS_GATE_BY_TAG.get(tag, {}).get(tuple(E2E_PLANES), 1e-6 if tuple(E2E_PLANES) in ((2, 2), (3, 3)) else 1e-3)
    worst_cos, worst_k, worst_f32 = 0.0, "", 0.0
    for k, got, t64, t32 in items:
        miss, miss32 = 1.0 - cosine(got, t64), 1.0 - cosine(t32, t64)
        if miss - 4.0 * miss32 > worst_cos:
            worst_cos, worst_k, worst_f32 = miss - 4.0 * miss32, k, miss32
    okc = bool(worst_cos <= cos_gate)
    RESULTS.append((f"{tag} MASKED grads: 1 - cosine with float64 [{worst_k}]", worst_cos, cos_gate, okc))
    print(f"{'ok  ' if okc else 'FAIL'} {tag} gradient directions: worst 1 - cos(g_gpu, g_f64) beyond the fp32 oracle's own = {worst_cos:.2e} "
          f"({worst_k}; fp32 oracle {worst_f32:.1e}); gate {cos_gate:.0e}", flush=True)
    # informational: the largest plain error among well-conditioned tensors (fp32 oracle within 1e-5 of float64)
    wc = [(util.relerr(got, t64), k) for k, got, t64, t32 in items if util.relerr(t32, t64) < 1e-5]
    if wc:
        e, k = max(wc)
        floor = WELL_FLOOR_BY_TAG.get(tag, {}).get(tuple(E2E_PLANES), floor)
        RESULTS.append((f"{tag} MASKED worst well-conditioned tensor [{k}]", e, floor, bool(e <= floor)))
        print(f"{'ok  ' if e <= floor else 'FAIL'} {tag} worst error among the {len(wc)} well-conditioned tensors: {k} {e:.2e} (floor {floor:.0e})")
    flip_gate = 2e-4 if E2E_PLANES[0] in (2, 3) else 5e-3
    for pre, (d, t) in (flips or {}).items():     # how many ReLU decisions differ from the fp32 oracle's own
        frac = d / max(t, 1)
        RESULTS.append((f"{tag} ReLU decisions differing from fp32 [{pre}]", frac, flip_gate, frac <= flip_gate))
        print(f"{'ok  ' if frac <= flip_gate else 'FAIL'} {tag} {pre}: {int(d)} of {t} ReLU decisions differ from the fp32 oracle ({frac:.2e})")


def _gpu_masks(keep, precision):
    """ReLU decisions of the coarse / fine / noise MLP evaluations of the last forward (the model's hooks.keep)."""
    pf, pb = precision.fwd, precision.bwd
    f16 = pf == ops.PLANES_F16
    sp = ops.nplanes(ops.stash_code(pf, pb))
    out = {"mlp_coarse": util.stash_masks(ops.NET_NERF, sp, keep["P_c"], keep["stash_c"], f16=f16)}
    if keep.get("stash_f") is not None:
        out["mlp_fine"] = util.stash_masks(ops.NET_NERF, sp, keep["P_f"], keep["stash_f"], f16=f16)
    if keep.get("stash_noise") is not None:
        pn = precision.noise()
        out["mlp_noise_coarse"] = util.stash_masks(ops.NET_NOISE, ops.nplanes(ops.stash_code(pn.fwd, pn.bwd)), keep["P_noise"],
                                                   keep["stash_noise"], f16=False)
    return out


def live_vs_dense(tag, net, rays, dense, dense_rays):
    """The product's live-point backward against the dense one on the same forward: every tensor to the order of the fp32 sums.
    Both add the same non-zero terms (a dead point's row is exact zeros) but group them differently -- 32 live points per
    weight-gradient stage instead of ~16 live + ~16 dead -- so a tensor that is a cancelling sum (the alpha bias, the RBK heads:
    the tensors on which the fp32 oracle itself is off by 1e-4 .. 1e-3 against float64) moves by its condition number times 2^-24
    per partial sum; gates: all gradients together 1e-5 of their norm, no tensor beyond 2e-3 of its largest entry."""
    num = den = 0.0
    worst, wk = 0.0, ""
    items = [(k, v.grad, dense[k]) for k, v in net.named_parameters() if v.grad is not None] + [("d rays", rays.grad, dense_rays)]
    for k, a, b in items:
        e = util.relerr(a, b)
        if e > worst:
            worst, wk = e, k
        num += float((a.double() - b.double()).pow(2).sum())
        den += float(b.double().pow(2).sum())
    l2 = (num / max(den, 1e-300)) ** 0.5
    RESULTS.append((f"{tag} live-point backward = dense backward (all gradients, L2)", l2, 1e-5, bool(l2 <= 1e-5)))
    RESULTS.append((f"{tag} live-point backward = dense backward (worst tensor [{wk}])", worst, 2e-3, bool(worst <= 2e-3)))
    print(f"{'ok  ' if l2 <= 1e-5 and worst <= 2e-3 else 'FAIL'} {tag}: live-point backward against the dense one: L2 {l2:.1e}, worst tensor {wk} {worst:.1e}", flush=True)


def t_train_e2e():
    from lush_nerf_amd import model as M
    import argparse
    for name in ("train_naive_sharp", "train_kernel_sharp", "train_kernel_default"):
        g = util.golden(name)
        n, Ns, Ni, naive, sharp, seed, allk = (int(x) for x in g["meta"])
        args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                                  N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                                  rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                                  render_rmnearplane=80)
        rbk = M.RBK(util.NUM_IMG, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4)
        prec = ops.Precision(*E2E_PLANES)
        net = M.NeRFAll(args, rbk, precision=prec)
        rbk_scale = 1.0 if naive else 2.0e4
        M.load_reference_weights(net, synth.all_weights(util.NUM_IMG, seed, sharp=bool(sharp), rbk_scale=rbk_scale))
        net = net.to(dev).train()
        b = batch_of(n, seed)
        K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
        cpu_draws = util.tdraws(n * (1 if naive else 5), Ns, Ni, seed)
        draws = {k: v.to(dev) for k, v in cpu_draws.items()}
        rays = gpu(b["rays"]).requires_grad_(True)
        keep = net.hooks.keep = {}
        try:
            out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info={"images_idx": gpu(b["images_idx"])}, retraw=True,
                      force_naive=bool(naive), allkernel=bool(allk), kernel_pixel=gpu(b["fq_mask"]), perturb=1.,
                      N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1.,
                      inference=False, near=0., far=1., draws=draws)
        finally:
            net.hooks.keep = None
        loss = ops.TrainLoss.apply(out[0], out[1], gpu(b["target"]))
        loss.backward()
        if ops.live_backward(prec, None):
            # the PRODUCT's march keeps no stash and runs its backward on the live points only (include/lush_march.h "Live points");
            # the run above kept every point's stash for the ReLU decisions (hooks.keep).  Same forward, product backward: the
            # outputs must be the same bit for bit, the gradients equal up to the order of the fp32 atomics -- and THESE are the
            # gradients the oracle checks below see.
            dense = {k: (None if v.grad is None else v.grad.clone()) for k, v in net.named_parameters()}
            dense_rays, dense_out = rays.grad.clone(), [o.detach().clone() for o in (out[0], out[1], out[3])]
            net.zero_grad(set_to_none=True)
            rays.grad = None
            out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info={"images_idx": gpu(b["images_idx"])}, retraw=True,
                      force_naive=bool(naive), allkernel=bool(allk), kernel_pixel=gpu(b["fq_mask"]), perturb=1.,
                      N_importance=Ni, N_samples=Ns, use_viewdirs=True, white_bkgd=False, raw_noise_std=1.,
                      inference=False, near=0., far=1., draws=draws)
            loss = ops.TrainLoss.apply(out[0], out[1], gpu(b["target"]))
            loss.backward()
            same = all(torch.equal(a, o.detach()) for a, o in zip(dense_out, (out[0], out[1], out[3])))
            RESULTS.append((f"train {name} live-point march: outputs bit-identical to the dense march", 0. if same else 1., 0, same))
            live_vs_dense(f"train {name}", net, rays, dense, dense_rays)
        rep(f"train {name} rgb_blur", out[0], g["rgb_blur"], 1e-4)
        rep(f"train {name} rgb0_blur", out[1], g["rgb0_blur"], 1e-4)
        rep(f"train {name} noise", out[3], g["noise"], 1e-4)
        rep(f"train {name} loss", loss.reshape(1), g["loss"].reshape(1), 1e-4)
        fw = net.read_faults()
        RESULTS.append((f"train {name} fault word", float(fw), 0, fw == 0))
        sd = dict(net.named_parameters())
        grads = {}
        for k, v in sd.items():
            ck = k
            if k.startswith("blur_kernel_net.RBK."):
                ck = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
            elif k.startswith("blur_kernel_net.view_embed_layer."):
                ck = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
            grads[ck] = v.grad
        none_ref = set(str(x) for x in g["grad_none"])
        none_got = set(k for k, v in grads.items() if v is None)
        ok = none_ref == none_got
        RESULTS.append((f"train {name} grad-None set", 0. if ok else 1., 0, ok))
        print("ok  " if ok else "FAIL", f"train {name} grad None set", sorted(none_ref ^ none_got)[:6])
        # (1) PRIMARY gradient gate (masked_grad_check above).  Any exception or NaN fails.
        def run_oracle(dt, naive=naive, sharp=sharp, seed=seed, allk=allk, Ns=Ns, Ni=Ni, b=b, cpu_draws=cpu_draws, rbk_scale=rbk_scale):
            p = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=bool(sharp), rbk_scale=rbk_scale).items()}
            rays_c = b["rays"].to(dt).clone().requires_grad_(True)
            ref = O.forward_train(p, H, W, F, rays_c, b["images_idx"], Ns, Ni, force_naive=bool(naive), allkernel=bool(allk),
                                  kernel_pixel=b["fq_mask"], draws={k: v.to(dt) for k, v in cpu_draws.items()})
            O.train_loss(ref[0], ref[1], b["target"].to(dt)).backward()
            return p, {"grad_rays": rays_c}
        masked_grad_check(f"train {name}", run_oracle, grads, {"grad_rays": rays.grad}, keep, prec)
        # (2) SECONDARY: against the reference fixture, un-masked (includes the discrete effect of flipped kinks)
        worst = util.check_grads({k: v for k, v in grads.items() if v is not None}, g, 1e9)
        wk = max(worst, key=worst.get)
        sec = 3e-2 if E2E_PLANES[0] in (2, 3) else 1e-1
        RESULTS.append((f"train {name} worst grad vs fixture [{wk}]", worst[wk], sec, bool(worst[wk] <= sec)))
        print(f"{'ok  ' if worst[wk] <= sec else 'FAIL'} train {name} worst grad vs reference fixture {wk}: {worst[wk]:.3e}")
        for k in sorted(worst, key=worst.get, reverse=True)[:3]:
            print(f"       {k}: {worst[k]:.3e}")
        if g["grad_rays"].size and rays.grad is not None:
            rep(f"train {name} grad_rays vs fixture", rays.grad, g["grad_rays"], sec)


def t_train_bench_regime(n=512, seed=21, Ns=64, Ni=64, trained_like=False, min_tiles=2048):
    """The regime bench.py runs the chain kernels in: MANY 128-point tiles per persistent workgroup, so the weight stream
    wraps into the next tile (mlp_chain_*_half_kernel: min(n_tiles, 2 n_cu) workgroups; the 512-register kernels: n_cu)
    and dw_group_kernel walks many slices.  N_rand 512 with the blur kernel on = 2 560 marched rays = 327 680 fine points
    = 2 560 tiles (5 per workgroup for the half-row kernels, 10 for the one-workgroup ones).  No reference fixture exists
    at this size; the oracle (pinned to the fixtures by tests/test_oracle_golden.py) is evaluated here: outputs against
    its plain fp32 run at the north-star bound, gradients through the masked float64 / fp32 pair of masked_grad_check
    (models/lushnerf.py:481-583, 630-654; run_lushnerf.py:652-661).
    (n, Ns, Ni) = (256, 128, 128) is BASELINE config 5's sampling at the same point count (1 280 marched rays x 256 =
    327 680 fine points): composite_*_kernel<4>, sample_merge at 128 + 128, ray_grad_reduce over 256 samples and the
    chain / weight-gradient kernels with 256 consecutive points per ray.
    trained_like (round 6): the same step from synth.all_weights(trained_like=True) -- the density field bench.py's `trained_like`
    workload starts from (empty space and surfaces behind a x3000 density head; calibrated for seed 0) -- instead of the sharp
    fixture field, in which the first sample of every ray is opaque."""
    prec = ops.Precision(*E2E_PLANES)
    rbk_scale = 2.0e4
    sharp = not trained_like
    net = _nerf_all(Ni, seed, sharp=sharp, precision=prec, rbk_scale=rbk_scale, trained_like=trained_like)
    b = batch_of(n, seed)
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    cpu_draws = util.tdraws(n * 5, Ns, Ni, seed)
    draws = {k: v.to(dev) for k, v in cpu_draws.items()}
    rays = gpu(b["rays"]).requires_grad_(True)
    keep = net.hooks.keep = {}
    try:
        out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info={"images_idx": gpu(b["images_idx"])}, retraw=True,
                  force_naive=False, allkernel=True, kernel_pixel=gpu(b["fq_mask"]), perturb=1., N_importance=Ni, N_samples=Ns,
                  use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False, near=0., far=1., draws=draws)
    finally:
        net.hooks.keep = None
    loss = ops.TrainLoss.apply(out[0], out[1], gpu(b["target"]))
    loss.backward()
    live_frac = None
    if ops.live_backward(prec, None):      # the product's march: backward on the live points only (see t_train_e2e); ITS gradients go to the oracle
        dense = {k: (None if v.grad is None else v.grad.clone()) for k, v in net.named_parameters()}
        dense_rays, dense_out = rays.grad.clone(), [o.detach().clone() for o in (out[0], out[1], out[3], out[5])]
        net.zero_grad(set_to_none=True)
        rays.grad = None
        out = net(H, W, K, chunk=1 << 20, rays=rays, rays_info={"images_idx": gpu(b["images_idx"])}, retraw=True,
                  force_naive=False, allkernel=True, kernel_pixel=gpu(b["fq_mask"]), perturb=1., N_importance=Ni, N_samples=Ns,
                  use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False, near=0., far=1., draws=draws)
        loss = ops.TrainLoss.apply(out[0], out[1], gpu(b["target"]))
        loss.backward()
        same = all(torch.equal(a, o.detach()) for a, o in zip(dense_out, (out[0], out[1], out[3], out[5])))
        RESULTS.append(("bench regime live-point march: outputs bit-identical to the dense march", 0. if same else 1., 0, same))
        live_vs_dense("bench regime", net, rays, dense, dense_rays)
    fw = net.read_faults()
    RESULTS.append(("bench regime fault word", float(fw), 0, fw == 0))
    tiles = keep["P_f"] // 128
    info = f"{keep['P_f']} fine points = {tiles} tiles"
    RESULTS.append((f"bench regime is multi-tile ({info})", float(tiles), float(min_tiles), tiles >= min_tiles))
    if live_frac is None and trained_like and "live_share" in keep:
        live_frac = keep["live_share"]
    # (1) outputs against the plain fp32 oracle (its own ReLU decisions)
    with torch.no_grad():
        p = util.params(seed, sharp=sharp, rbk_scale=rbk_scale, trained_like=trained_like)
        ref = O.forward_train(p, H, W, F, b["rays"], b["images_idx"], Ns, Ni, force_naive=False, allkernel=True,
                              kernel_pixel=b["fq_mask"], draws=cpu_draws)
        ref_loss = O.train_loss(ref[0], ref[1], b["target"])
    tagname = "trained-like regime" if trained_like else ("bench regime" if (Ns, Ni) == (64, 64) else f"bench regime {Ns}+{Ni}")
    rep("bench regime rgb_blur", out[0], ref[0], 1e-4)
    rep("bench regime rgb0_blur", out[1], ref[1], 1e-4)
    rep("bench regime noise", out[3], ref[3], 1e-4)
    rep("bench regime rgb (sharp)", out[5], ref[5], 1e-4)
    rep("bench regime loss", loss.reshape(1), ref_loss.reshape(1), 1e-4)
    # (2) gradients: the masked float64 / fp32 oracle pair
    def run_oracle(dt):
        pp = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=sharp, rbk_scale=rbk_scale, trained_like=trained_like).items()}
        rays_c = b["rays"].to(dt).clone().requires_grad_(True)
        r = O.forward_train(pp, H, W, F, rays_c, b["images_idx"], Ns, Ni, force_naive=False, allkernel=True,
                            kernel_pixel=b["fq_mask"], draws={k: v.to(dt) for k, v in cpu_draws.items()})
        O.train_loss(r[0], r[1], b["target"].to(dt)).backward()
        return pp, {"grad_rays": rays_c}
    masked_grad_check(tagname, run_oracle, _canon_grads(net), {"grad_rays": rays.grad}, keep, prec)


def t_train_c1():
    """BASELINE config 1 as bench.py times it (Trainer.step_coarse_only: render_infer at N_rand 256, 32 + 0, naive, ONE summed
    loss gradient, Adam), eagerly AND as the replayed HIP graph, against the reference fixture `train_c1` and the oracle
    (models/lushnerf.py:679-763 -> :354-479; run_lushnerf.py:652-661 with rgb0 = rgb)."""
    from lush_nerf_amd.trainer import Trainer
    from tests.test_oracle_golden import oracle_c1_step
    g = util.golden("train_c1")
    n, Ns, Ni, seed = (int(x) for x in g["meta"])
    prec = ops.Precision(*E2E_PLANES)

    def fresh():
        net = _nerf_all(0, seed, sharp=True, precision=prec)
        return net, Trainer(net, H, W, F, Ns, 0, kernel_start_iter=1 << 30)
    b = batch_of(n, seed)
    gb = {"rays": gpu(b["rays"]), "target": gpu(b["target"])}
    cpu_draws = util.tdraws(n, Ns, 0, seed)
    w0 = {k: torch.from_numpy(v) for k, v in synth.all_weights(util.NUM_IMG, seed, sharp=True).items() if not k.startswith("mlp_fine.")}
    # (1) eager, explicit draws: outputs, loss, gradient-None set, gradients (masked float64 / fp32 pair + the fixture), Adam
    net, tr = fresh()
    keep = net.hooks.keep = {}
    try:
        loss, tm = tr.step_coarse_only(gb, draws={k: v.to(dev) for k, v in cpu_draws.items()})
    finally:
        net.hooks.keep = None
    rep("c1 step rgb (tone-mapped)", tm, g["rgb_tm"], 1e-4)
    rep("c1 step loss", loss.reshape(1), g["loss"].reshape(1), 1e-4)
    fw = net.read_faults()
    RESULTS.append(("c1 step fault word", float(fw), 0, fw == 0))
    grads = {k: (None if v is None else v.clone()) for k, v in _canon_grads(net).items()}
    coarse = {k: v for k, v in grads.items() if k.startswith("mlp_coarse.")}
    # the trainer's flat gradient gives every parameter a .grad view; what this step must leave untouched (zero) is every
    # tensor the reference leaves with grad = None (RBK, noise MLP: the noise branch is detached from the loss here)
    none_ref = set(str(x) for x in g["grad_none"])
    touched = set(k for k, v in grads.items() if v is not None and float(v.abs().max()) > 0)
    ok = touched == set(grads) - none_ref
    RESULTS.append(("c1 step: exactly the reference's grad-None tensors stay zero", 0. if ok else 1., 0, ok))
    print("ok  " if ok else "FAIL", "c1 step grad-None set", sorted(touched ^ (set(grads) - none_ref))[:6])

    def run_oracle(dt):
        p = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=True).items() if not k.startswith("mlp_fine.")}
        _, _, _, lo = oracle_c1_step(p, b["rays"].to(dt), b["target"].to(dt), Ns, {k: v.to(dt) for k, v in cpu_draws.items()})
        lo.backward()
        return p, {}
    masked_grad_check("c1 step", run_oracle, coarse, {}, keep, prec)
    worst = util.check_grads(coarse, g, 1e9)
    wk = max(worst, key=worst.get)
    sec = 3e-2 if E2E_PLANES[0] in (2, 3) else 1e-1
    RESULTS.append((f"c1 step worst grad vs fixture [{wk}]", worst[wk], sec, bool(worst[wk] <= sec)))
    print(f"{'ok  ' if worst[wk] <= sec else 'FAIL'} c1 step worst grad vs reference fixture {wk}: {worst[wk]:.3e}")
    # Adam's first step on these gradients: p - lr * g / (|g| + eps) (torch.optim.Adam, bias corrections cancel at step 1)
    a0, a1 = tr.flat.segments[0]
    flat0 = torch.cat([gpu(w0[k]).reshape(-1) for k in _flat_order(net)[0]])
    gflat = tr.flat.grad[a0:a1]
    want = flat0 - 5e-4 * gflat / (gflat.abs() + 1e-8)
    rep("c1 step Adam update of the coarse network", tr.flat.param[a0:a1], want, 1e-6)
    rest_same = torch.equal(tr.flat.param[a1:], torch.cat([gpu(w0[k]).reshape(-1) for k in _flat_order(net)[1]]))
    RESULTS.append(("c1 step leaves RBK / noise parameters alone", 0. if rest_same else 1., 0, rest_same))
    # (2) the same step as a replayed HIP graph (device step state: draw counter, rate, Adam count) against an eager trainer fed
    # the draws the graph's Philox call makes (same seed / offset): losses, colours, gradients, parameters after 3 steps
    torch.manual_seed(1234)
    net_g, tr_g = fresh()
    net_e, tr_e = fresh()
    batches = [{"rays": gpu(batch_of(n, seed + 1 + s)["rays"]), "target": gpu(batch_of(n, seed + 1 + s)["target"])} for s in range(4)]
    tr_g.step_coarse_only(batches[0])              # one eager step first (lazy initialisation outside the capture)
    d0 = ops.march_draws(n, Ns, 0, 1., 1., dev, offset=1)
    tr_e.step_coarse_only(batches[0], draws=d0)
    net_e.hooks.draw_offset = net_g.hooks.draw_offset
    replay, static = tr_g.capture_coarse_only(batches[1])
    for s in (1, 2, 3):
        # every replay against ONE eager step from the same state (two free-running trainers drift apart: fp32 atomics order the
        # gradient sums differently, and Adam's g / (|g| + eps) turns a sign flip of a ~0 entry into a 2 lr step)
        for dst, src in ((tr_e.flat.param, tr_g.flat.param), (tr_e.m, tr_g.m), (tr_e.v, tr_g.v)):
            dst.copy_(src)
        for k in static:
            static[k].copy_(batches[s][k])
        off = net_g.hooks.draw_offset + 1
        replay()
        ds = ops.march_draws(n, Ns, 0, 1., 1., dev, offset=off)
        le, tme = tr_e.step_coarse_only(batches[s], draws=ds)
        rep(f"c1 graph replay {s}: loss vs eager", replay.loss.reshape(1), le.reshape(1), 2e-6)
        rep(f"c1 graph replay {s}: colours vs eager", replay.tm, tme, 2e-6)
        rep(f"c1 graph replay {s}: gradient vs eager", tr_g.flat.grad, tr_e.flat.grad, 2e-5)
        dp = (tr_g.flat.param - tr_e.flat.param).abs()
        far = float((dp > 1e-6 * float(tr_e.flat.param.abs().max())).float().mean())
        okp = far <= 1e-4 and float(dp.max()) <= 2.1 * 5e-4
        RESULTS.append((f"c1 graph replay {s}: parameters vs eager (share beyond 1e-6; sign-sensitive Adam entries)", far, 1e-4, okp))
        print(f"{'ok  ' if okp else 'FAIL'} c1 graph replay {s}: parameters after the step: {far:.1e} of the entries beyond 1e-6, largest difference {float(dp.max()):.1e}", flush=True)
    cnt_ok = tr_g.steps == tr_e.steps and tr_g.global_step == tr_e.global_step == 4
    RESULTS.append(("c1 graph: host counters advance like the eager step's", 0. if cnt_ok else 1., 0, cnt_ok))


class KeepAll(dict):
    """ops.Hooks.keep that remembers EVERY march of a step (the plain dict keeps the last one's stashes): the combined step of
    the consistency branch marches twice through the same two networks."""

    def __init__(self):
        super().__init__()
        self.marches = []

    def update(self, *a, **kw):
        if "stash_c" in kw:
            self.marches.append(dict(kw))
        super().update(*a, **kw)


def _masks_of_all_marches(keep, precision):
    """ReLU decisions of every coarse / fine / noise evaluation of a step, rows in call order (util.masked_oracle consumes them
    with a per-prefix cursor)."""
    pf, pb = precision.fwd, precision.bwd
    f16 = pf == ops.PLANES_F16
    sp = ops.nplanes(ops.stash_code(pf, pb))
    per = {"mlp_coarse": [], "mlp_fine": []}
    for m in keep.marches:
        per["mlp_coarse"].append(util.stash_masks(ops.NET_NERF, sp, m["P_c"], m["stash_c"], f16=f16))
        if m.get("stash_f") is not None:
            per["mlp_fine"].append(util.stash_masks(ops.NET_NERF, sp, m["P_f"], m["stash_f"], f16=f16))
    out = {k: [torch.cat([c[l] for c in v], 0) for l in range(len(v[0]))] for k, v in per.items() if v}
    if keep.get("stash_noise") is not None:
        pn = precision.noise()
        out["mlp_noise_coarse"] = util.stash_masks(ops.NET_NOISE, ops.nplanes(ops.stash_code(pn.fwd, pn.bwd)), keep["P_noise"],
                                                   keep["stash_noise"], f16=False)
    return out


def t_consist_step():
    """Trainer.step(batch, i, consist=...) -- the combined step after noisenerf_start_iter (run_lushnerf.py:625-661): the ray
    batch's march AND the aligned-pixel march accumulate into ONE flat gradient, loss = image terms + 1e-2 * loss_rgb for
    i > noisenerf_start_iter, + 0 at i == noisenerf_start_iter (computed, not added).  Against the reference fixture
    `train_consist` and the masked float64 / fp32 oracle; the `>` / `>=` edge; step_graph's fall-back."""
    from lush_nerf_amd.trainer import Trainer
    from tests.test_oracle_golden import consist_step_inputs, oracle_consist_step
    g = util.golden("train_consist")
    n, Ns, Ni, seed, V, ns, anchor, st, am, cm = consist_step_inputs(g)
    prec = ops.Precision(*E2E_PLANES)
    rbk_scale, START = 2.0e4, 5
    b = batch_of(n, seed)
    gb = {k: gpu(v) for k, v in b.items()}
    poses = torch.from_numpy(synth.poses(V, seed))
    consist = {"poses": gpu(poses), "images_idx": torch.arange(V), "Align_matrix": {anchor: gpu(am)}, "Align_mask": {anchor: gpu(cm)},
               "anchor_pose": anchor, "samples": st}
    cpu_draws = util.tdraws(n * 5, Ns, Ni, seed)
    draws = {k: v.to(dev) for k, v in cpu_draws.items()}

    def fresh():
        net = _nerf_all(Ni, seed, sharp=True, precision=prec, rbk_scale=rbk_scale)
        return net, Trainer(net, H, W, F, Ns, Ni, kernel_start_iter=0, allkernel_start_iter=0, noisenerf_start_iter=START)
    # (1) i > noisenerf_start_iter: weight 1e-2
    net, tr = fresh()
    keep = net.hooks.keep = KeepAll()
    try:
        loss = tr.step(gb, START + 1, draws=draws, consist=consist)
    finally:
        net.hooks.keep = None
    rep("consist step loss (image terms + 1e-2 loss_rgb)", loss.reshape(1), g["loss"].reshape(1), 1e-4)
    fw = net.read_faults()
    RESULTS.append(("consist step fault word", float(fw), 0, fw == 0))
    two = len(keep.marches) == 2
    RESULTS.append(("consist step marched twice (ray batch, aligned pixels)", float(len(keep.marches)), 2, two))
    grads = {k: v.clone() for k, v in _canon_grads(net).items()}
    none_ref = set(str(x) for x in g["grad_none"])
    live = {k: v for k, v in grads.items() if k not in none_ref}
    dead_zero = all(float(grads[k].abs().max()) == 0.0 for k in none_ref if k in grads)
    RESULTS.append(("consist step: the reference's grad-None tensors stay zero", 0. if dead_zero else 1., 0, dead_zero))

    def run_oracle(dt):
        p = {k: v.to(dt).requires_grad_(True) for k, v in util.params(seed, sharp=True, rbk_scale=rbk_scale).items()}
        out = oracle_consist_step(p, b, Ns, Ni, {k: v.to(dt) for k, v in cpu_draws.items()}, poses, am, cm, st, 1e-2, dt)
        out[-1].backward()
        return p, {}
    # (masked_grad_check reads the masks through _gpu_masks(keep, prec): hand it every march's rows)
    global _gpu_masks
    saved = _gpu_masks
    _gpu_masks = _masks_of_all_marches
    try:
        masked_grad_check("consist step", run_oracle, live, {}, keep, prec)
    finally:
        _gpu_masks = saved
    worst = util.check_grads(live, g, 1e9)
    wk = max(worst, key=worst.get)
    sec = 3e-2 if E2E_PLANES[0] in (2, 3) else 1e-1
    RESULTS.append((f"consist step worst grad vs fixture [{wk}]", worst[wk], sec, bool(worst[wk] <= sec)))
    print(f"{'ok  ' if worst[wk] <= sec else 'FAIL'} consist step worst grad vs reference fixture {wk}: {worst[wk]:.3e}")
    # the consistency term must actually reach the gradient: against the same step without it
    net0, tr0 = fresh()
    loss0 = tr0.step(gb, START + 1, draws=draws)
    rep("plain step loss (image terms)", loss0.reshape(1), g["loss_img"].reshape(1), 1e-4)
    moved = util.relerr(tr.flat.grad, tr0.flat.grad)
    RESULTS.append(("consist step: the 1e-2 loss_rgb term changes the gradient", moved, 1e-4, moved > 1e-4))
    print(f"{'ok  ' if moved > 1e-4 else 'FAIL'} consist step gradient differs from the plain step's by {moved:.2e}")
    # (2) i == noisenerf_start_iter: computed, weight 0 (run_lushnerf.py:629 `>=`, :658 `>`): the plain step's loss and gradient
    net1, tr1 = fresh()
    loss1 = tr1.step(gb, START, draws=draws, consist=consist)
    rep("consist step at i == noisenerf_start_iter: loss = image terms", loss1.reshape(1), loss0.reshape(1), 1e-6)
    rep("consist step at i == noisenerf_start_iter: gradient = plain step's", tr1.flat.grad, tr0.flat.grad, 2e-5)
    rep("consist step at i == noisenerf_start_iter: parameters = plain step's", tr1.flat.param, tr0.flat.param, 1e-6)
    # (3) step_graph with the consistency branch falls back to step() and keeps the term (device draws, same seed and offsets)
    torch.manual_seed(4321)
    net2, tr2 = fresh()
    net3, tr3 = fresh()
    for i in range(2):       # the two plain steps after which step_graph would capture
        tr2.step_graph(gb, i)
        tr3.step(gb, i)
    for dst, src in ((tr3.flat.param, tr2.flat.param), (tr3.m, tr2.m), (tr3.v, tr2.v)):      # (one common state: two free-running
        dst.copy_(src)                                                                        # trainers drift by Adam's sign-sensitive entries)
    l2 = tr2.step_graph(gb, START + 1, consist=consist)
    l3 = tr3.step(gb, START + 1, consist=consist)
    rep("step_graph(consist=...) falls back: loss", l2.reshape(1), l3.reshape(1), 2e-6)
    rep("step_graph(consist=...) falls back: gradient", tr2.flat.grad, tr3.flat.grad, 2e-5)
    nograph = tr2._graph is None
    RESULTS.append(("step_graph(consist=...) captured nothing", 0. if nograph else 1., 0, nograph))


def _flat_order(net):
    """Canonical names of the trainer's three Adam segments (trainer.adam_segments), in flat-buffer order."""
    from lush_nerf_amd.trainer import adam_segments
    name_of = {}
    for k, v in net.named_parameters():
        ck = k
        if k.startswith("blur_kernel_net.RBK."):
            ck = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        elif k.startswith("blur_kernel_net.view_embed_layer."):
            ck = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        name_of.setdefault(id(v), ck)
    segs = adam_segments(net)
    seen, first, rest = set(), [], []
    for si, seg in enumerate(segs):
        for p_ in seg:
            if id(p_) in seen:
                continue
            seen.add(id(p_))
            (first if si == 0 else rest).append(name_of[id(p_)])
    return first, rest


if __name__ == "__main__":
    lib.load()
    print("device:", torch.cuda.get_device_name(0))
    only = sys.argv[1:]
    for fn in (t_zgrid_pack, t_gen_rays, t_composite, t_sample, t_mlp_fwd, t_mlp_ragged, t_mlp_bwd, t_rbk, t_warp_ndc, t_mix, t_blur_mix, t_march_e2e, t_train_e2e,
               t_lindisp_white, t_eval_forward, t_consistency, t_faults, t_draws, t_train_bench_regime, t_train_c1, t_consist_step):
        if not only or fn.__name__ in only:
            section(fn)
    bad = [r for r in RESULTS if not r[3]]
    print(f"\n{len(RESULTS) - len(bad)} ok, {len(bad)} failing")
    for r in bad:
        print("  FAIL", r[0], f"{r[1]:.3e}")
    sys.exit(1 if bad else 0)
