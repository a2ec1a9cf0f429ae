#!/usr/bin/env python3
"""Worker of test_ab_switches_agree (needs a GPU): one forward + backward of the fine MLP in the mode given by
LUSH_PLANES on fixed synthetic inputs; writes raw outputs, d(point) and every parameter gradient to argv[1].
The kernel variant is the lib.VARIANT_* bit mask in LUSH_VARIANT (read HERE, by the test worker: the library itself takes
the variant as an argument and never looks at the environment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lush_nerf_amd import ops, synth
from oracle import lush_oracle as O       # (test infrastructure: ray packing of the synthetic batch only)

dev = torch.device("cuda:0")
R, S = int(os.environ.get("LUSH_AB_R", 96)), int(os.environ.get("LUSH_AB_S", 100))
pf, pb = ops.parse_planes(os.environ.get("LUSH_PLANES", "h,h"))
variant = int(os.environ.get("LUSH_VARIANT", "0"))
w = synth.all_weights(30, 3, sharp=True)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
b = synth.ray_batch(R, 5)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
g = torch.Generator().manual_seed(11)
z = torch.sort(torch.rand(R, S, generator=g), -1)[0].to(dev)
draw = (torch.randn(R * S, 4, generator=g) * 1e-2).to(dev)
pk = ops.mlp_pack(0, pf, tens)
pkb = pk if pb == pf else ops.mlp_pack(0, pb, tens)
raw, stash = ops.mlp_forward(0, pf, tens, pk, batch, z, True, ops.stash_code(pf, pb), variant)
grads, dpts = ops.mlp_backward(0, ops.stash_code(pf, pb), pb, tens, pkb, batch, z, draw, stash, variant=variant)
torch.cuda.synchronize()
np.savez(sys.argv[1], raw=raw.cpu().numpy(), dpts=dpts.cpu().numpy(), **{f"g{i}": t.cpu().numpy() for i, t in enumerate(grads)})
