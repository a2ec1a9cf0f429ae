#!/usr/bin/env python3
"""Developer tool (needs a GPU and a -DLUSH_PROF build named by LUSH_SO): cycle budget of mlp_wide_fwd_kernel, block 0 / wave 0.
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
from oracle import lush_oracle as O

dev = torch.device("cuda:0")
R, S = int(os.environ.get("R", 20480)), int(os.environ.get("S", 128))
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
b = synth.ray_batch(R, 1)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
L = lib.load()
pk = ops.mlp_pack(0, ops.PLANES_F16, tens)
for stash in (True, False):
    for _ in range(2):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.mlp_forward(0, ops.PLANES_F16, tens, pk, batch, z, stash, ops.PLANES_F16 if stash else 0)
        e.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)(); L.lush_debug_prof_wide(out); v = list(out); ms = a.elapsed_time(e)
    tiles, npos = max(v[8], 1), max(v[7], 1)
    print(f"stash={stash} ms={ms:.3f} kernel_cycles={v[0]} -> {v[0]/ms/1e3:.0f} MHz; tiles={tiles}; per tile: total={v[0]/tiles:.0f} "
          f"pe+setup={v[1]/tiles:.0f} L0={v[2]/tiles:.0f} trunk={v[3]/tiles:.0f} tail={v[4]/tiles:.0f}; per position: "
          f"vmcnt wait={v[5]/npos:.0f} lgkm+barrier={v[6]/npos:.0f} (positions/tile={npos/tiles:.1f})")
