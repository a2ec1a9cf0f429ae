#!/usr/bin/env python3
"""Developer tool: cycle breakdown of the chain forward kernel (needs a LUSH_PROF build via LUSH_SO).  Not a test."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lush_nerf_amd import lib, ops, synth
if os.environ.get("LUSH_SO"):        # developer tool: a variant built by tools/build_variant.py
    lib.use_library(os.environ["LUSH_SO"])
from oracle import lush_oracle as O
dev = torch.device("cuda:0")
R, S = 20480, 128
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
b = synth.ray_batch(R, 1)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
L = lib.load()
L.lush_debug_prof.argtypes = [C.POINTER(C.c_ulonglong)]
for mode, stash in ((2, False), (2, True), (17, False)):
    pk = ops.mlp_pack(0, mode, tens)
    for _ in range(2):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.mlp_forward(0, mode, tens, pk, batch, z, stash, 1 if stash else 0); e.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.lush_debug_prof(out)
    ms = a.elapsed_time(e)
    v = list(out)
    print(f"mode={mode} stash={stash} ms={ms:.3f} kernel_cycles={v[0]} -> {v[0]/ms/1e3:.0f} MHz; pe={v[1]} trunk(l1..7)={v[2]} phases={v[3]} conv={v[4]}  per-tile: total={v[0]/80:.0f} phase/layer={v[3]/80/7:.0f} conv/layer={v[4]/80/7:.0f}")

# backward chain kernel
import ctypes as C
for nf, nb in ((2, 1), (2, 2)):
    pk = ops.mlp_pack(0, nf, tens); pkb = ops.mlp_pack(0, nb, tens)
    raw, stash = ops.mlp_forward(0, nf, tens, pk, batch, z, True, ops.stash_code(nf, nb))
    draw = torch.randn(R * S, 4, device=dev) * 1e-3
    dstash = torch.empty(L.lush_mlp_dstash_bytes(0, nb, R * S), dtype=torch.uint8, device=dev)
    dpts = torch.empty(R * S, 8, device=dev)
    st = lib.mlp_struct(tens, 8)
    for _ in range(2):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.call("lush_mlp_bwd_chain", 0, ops.stash_code(nf, nb), nb, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pkb), C.byref(st),
                 lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), lib.ptr(dpts), 0, ops._stream())
        e.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)(); L.lush_debug_prof(out); v = list(out); ms = a.elapsed_time(e)
    print(f"bwd planes={nb} ms={ms:.3f} kernel_cycles={v[0]} -> {v[0]/ms/1e3:.0f} MHz; per tile: total={v[0]/80:.0f} pe_bwd={v[1]/80:.0f} phase/layer={v[3]/80/7:.0f} conv/layer={v[4]/80/7:.0f}")
