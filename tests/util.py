"""Shared helpers for the tests (inputs are regenerated from seeds, never stored)."""
import os

import numpy as np
import torch

from lush_nerf_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H, W, FOCAL, NUM_IMG = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 30


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def params(seed, sharp=False, rbk_scale=1.0, requires_grad=False, device="cpu", trained_like=False):
    w = synth.all_weights(NUM_IMG, seed, sharp=sharp, rbk_scale=rbk_scale, trained_like=trained_like)
    return {k: torch.from_numpy(v.copy()).to(device).requires_grad_(requires_grad) for k, v in w.items()}


def tdraws(R, Ns, Ni, seed, device="cpu"):
    return {k: torch.from_numpy(v).to(device) for k, v in synth.draws(R, Ns, Ni, seed).items()}


def proj_vecs(name, numel):
    return synth.normal((8, numel), 1234, synth._stream("proj." + name)).astype(np.float64)


def relerr(a, b):
    """max|a-b| / max(|b|, tiny): the normalised max error used for parity gates."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def check_grads(named_grads, fixture, tol, skip_missing=False):
    """Compare a {canonical name -> grad tensor/None} dict with a train_* fixture."""
    worst = {}
    for key in fixture:
        if key.startswith("grad.") and not key.startswith("grad_"):
            k = key[5:]
            g = named_grads[k].detach().cpu().numpy()
            worst[k] = relerr(g, fixture[key])
        elif key.startswith("gradproj."):
            k = key[9:]
            g = named_grads[k].detach().cpu().numpy().reshape(-1).astype(np.float64)
            pr = proj_vecs(k, g.size) @ g
            nrm = float(fixture["gradnorm." + k])
            # projections of a vector of norm nrm onto N(0,1) vectors have scale nrm
            worst[k] = float(np.max(np.abs(pr - fixture[key])) / max(nrm, 1e-30))
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"gradient mismatch (normalised max err > {tol}): {bad}"
    return worst


def stash_planes(stash, off, ns, Ppad, cols, f16=False):
    """Decode a [ns][Ppad][cols] bf16-plane (or one fp16 plane) array of the forward stash into fp32."""
    sb = stash.cpu().numpy() if isinstance(stash, torch.Tensor) else stash
    raw = sb[off:off + ns * Ppad * cols * 2].tobytes()
    with np.errstate(invalid="ignore"):      # rows beyond the last point of a 64-point-tile kernel are never written
        if f16:
            return np.frombuffer(raw, dtype=np.float16).astype(np.float32).reshape(ns, Ppad, cols).sum(0)
        arr = np.frombuffer(raw, dtype=np.uint16)
        return (arr.astype(np.uint32) << 16).view(np.float32).reshape(ns, Ppad, cols).sum(0)


def stash_masks(net, ns, P, stash, f16=False):
    """ReLU decisions the GPU forward took: [h_0 > 0, ..., h_{NL-1} > 0, hv > 0] as float tensors."""
    import ctypes as C
    from lush_nerf_amd import lib
    off = (C.c_longlong * 16)()
    lib.call("lush_debug_stash_layout", net, ns, P, off)
    Ppad, HW, NL = off[12], off[14], off[15]
    sb = stash.cpu().numpy()
    out = [torch.from_numpy((stash_planes(sb, off[2 + l], ns, Ppad, HW, f16)[:P] > 0).astype(np.float32)) for l in range(NL)]
    out.append(torch.from_numpy((stash_planes(sb, off[11], ns, Ppad, HW // 2, f16)[:P] > 0).astype(np.float32)))
    return out


def nerf_mlp_masked(p, prefix, x, depth, masks):
    """oracle.nerf_mlp with every ReLU replaced by multiplication with a given 0/1 mask."""
    import torch.nn.functional as F
    pts, views = x[..., :63], x[..., 63:90]
    h = pts
    for i in range(depth):
        h = F.linear(h, p[f"{prefix}.pts_linears.{i}.weight"], p[f"{prefix}.pts_linears.{i}.bias"]) * masks[i]
        if i == 4:
            h = torch.cat([pts, h], -1)
    feat = F.linear(h, p[f"{prefix}.feature_linear.weight"], p[f"{prefix}.feature_linear.bias"])
    hv = F.linear(torch.cat([feat, views], -1), p[f"{prefix}.views_linears.0.weight"],
                  p[f"{prefix}.views_linears.0.bias"]) * masks[depth]
    rgb = F.linear(hv, p[f"{prefix}.rgb_linear.weight"], p[f"{prefix}.rgb_linear.bias"])
    alpha = F.linear(h, p[f"{prefix}.alpha_linear.weight"], p[f"{prefix}.alpha_linear.bias"])
    return torch.cat([rgb, alpha], -1)


def oracle_relu_decisions(p, prefix, x, depth):
    """The ReLU decisions the fp32 oracle takes on input rows x [P, 90]: [h_0 > 0, ..., h_{D-1} > 0, hv > 0]."""
    import torch.nn.functional as F
    pts, views = x[..., :63], x[..., 63:90]
    h, out = pts, []
    with torch.no_grad():
        for i in range(depth):
            h = F.relu(F.linear(h, p[f"{prefix}.pts_linears.{i}.weight"], p[f"{prefix}.pts_linears.{i}.bias"]))
            out.append((h > 0).float())
            if i == 4:
                h = torch.cat([pts, h], -1)
        feat = F.linear(h, p[f"{prefix}.feature_linear.weight"], p[f"{prefix}.feature_linear.bias"])
        hv = F.relu(F.linear(torch.cat([feat, views], -1), p[f"{prefix}.views_linears.0.weight"],
                             p[f"{prefix}.views_linears.0.bias"]))
        out.append((hv > 0).float())
    return out


class masked_oracle:
    """Context manager: inside it the oracle's MLPs (oracle.lush_oracle.nerf_mlp) use GIVEN ReLU decisions instead of
    their own -- {prefix: [mask per layer]} -- so gradients can be compared free of ReLU-kink flips.  Also records
    how many of the given decisions differ from the ones the fp32 oracle would have taken (`flips[prefix]`)."""

    def __init__(self, masks_by_prefix):
        self.masks = masks_by_prefix
        self.flips = {}
        self.cursor = {}          # a prefix may be evaluated in several calls (one per pose, per chunk): rows in call order

    def __enter__(self):
        from oracle import lush_oracle as O
        self.O, self.orig = O, O.nerf_mlp

        def patched(p, prefix, x, in_ch, in_ch_views, depth, skips=(4,), return_alpha=True):
            if prefix not in self.masks:
                return self.orig(p, prefix, x, in_ch, in_ch_views, depth, skips, return_alpha)
            c0 = self.cursor.get(prefix, 0)
            P = x.shape[0]
            m = [mm[c0:c0 + P] for mm in self.masks[prefix]]
            assert m[0].shape[0] == P, f"{prefix}: mask rows exhausted ({c0}+{P} of {self.masks[prefix][0].shape[0]})"
            self.cursor[prefix] = c0 + P
            own = oracle_relu_decisions({k: v.detach() for k, v in p.items() if k.startswith(prefix)}, prefix, x.detach(), depth)
            diff = sum(float((a != b).sum()) for a, b in zip(own, m))
            tot = sum(a.numel() for a in own)
            d0, t0 = self.flips.get(prefix, (0.0, 0))
            self.flips[prefix] = (d0 + diff, t0 + tot)
            out = nerf_mlp_masked(p, prefix, x, depth, m)
            return out if return_alpha else out[..., :3]
        O.nerf_mlp = patched
        return self

    def __exit__(self, *a):
        self.O.nerf_mlp = self.orig
