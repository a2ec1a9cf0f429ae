#!/usr/bin/env python3
"""Developer tool (needs a GPU and a -DLUSH_PROF build named by LUSH_SO): cycle budget of mlp_wide_bwd_kernel, block 0 / wave 0.
Since the kernels grew in round 3 the 16 cycle counters no longer fit their scalar registers: the compiler spills SGPRs, such a build
faulted on the GPU, and lib.build()'s ISA audit refuses it (isa_check rule R1: spill reload directly in front of an asm VMEM with a scalar base).  The figures
in DESIGN.md section 4 come from earlier builds of the forward in which the counters fitted; ablation builds (-DLUSH_ABL_*) are
the tool that still works."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lush_nerf_amd import lib, ops, synth
if os.environ.get("LUSH_SO"):        # developer tool: a variant built by tools/build_variant.py
    lib.use_library(os.environ["LUSH_SO"])
from oracle import lush_oracle as O       # (ray packing of the synthetic batch only)

dev = torch.device("cuda:0")
R, S = int(os.environ.get("R", 20480)), int(os.environ.get("S", 128))
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
b = synth.ray_batch(R, 1)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
draw = torch.randn(R * S, 4, device=dev) * 1e-3
L = lib.load()
H = ops.PLANES_F16
pk = ops.mlp_pack(0, H, tens)
raw, stash = ops.mlp_forward(0, H, tens, pk, batch, z, True, H)
dstash = torch.empty(L.lush_mlp_dstash_bytes(0, H, R * S), dtype=torch.uint8, device=dev)
dpts = torch.empty(R * S, 8, device=dev)
st = lib.mlp_struct(tens, 8)
for _ in range(3):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    lib.call("lush_mlp_bwd_chain", 0, H, H, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pk), C.byref(st), lib.ptr(draw), lib.ptr(stash),
             lib.ptr(dstash), lib.ptr(dpts), 0, ops._stream())
    e.record(); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)(); L.lush_debug_prof_wbwd(out); v = list(out); ms = a.elapsed_time(e)
tiles, npos = max(v[8], 1), max(v[7], 1)
print(f"ms={ms:.3f} (launch incl. grad_scale) kernel_cycles={v[0]} ; tiles={tiles}; per tile: total={v[0]/tiles:.0f} prefetch wait={v[1]/tiles:.0f} "
      f"barrier={v[2]/tiles:.0f} prologue={v[3]/tiles:.0f} body={v[4]/tiles:.0f} image+prefetch+dZ0 rows={v[9]/tiles:.0f} encoding={v[10]/tiles:.0f}; "
      f"per position: vmcnt wait={v[5]/npos:.0f} lgkm+barrier={v[6]/npos:.0f} (positions/tile={npos/tiles:.1f})")
