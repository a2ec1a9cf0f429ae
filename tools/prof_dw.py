#!/usr/bin/env python3
"""Developer tool (needs a GPU and a -DLUSH_PROF build named by LUSH_SO): where workgroup 0 of dw_group_kernel spends a launch.
  python tools/build_variant.py --out build/dwprof.so --flags=-DLUSH_PROF && LUSH_SO=build/dwprof.so python tools/prof_dw.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lush_nerf_amd import lib, ops, synth
lib.use_library(os.environ["LUSH_SO"])
from oracle import lush_oracle as O

dev = torch.device("cuda:0")
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
L = lib.load()
L.lush_debug_prof_dw.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
L.lush_debug_prof_dw_span.argtypes = [C.POINTER(C.c_ulonglong)]
H = ops.PLANES_F16
for R, S in ((20480, 64), (20480, 128)):
    b = synth.ray_batch(R, 1)
    batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
    z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
    draw = torch.randn(R * S, 4, device=dev) * 1e-3
    pk = ops.mlp_pack(0, H, tens)
    raw, stash = ops.mlp_forward(0, H, tens, pk, batch, z, True, H, 0)
    dstash = torch.empty(L.lush_mlp_dstash_bytes(0, H, R * S), dtype=torch.uint8, device=dev)
    grads = [torch.zeros_like(t) for t in tens]
    dpts = torch.empty(R * S, 8, device=dev)
    st, gs = lib.mlp_struct(tens, 8), lib.mlp_struct(grads, 8)
    lib.call("lush_mlp_bwd_chain", 0, H, H, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pk), C.byref(st), lib.ptr(draw), lib.ptr(stash),
             lib.ptr(dstash), lib.ptr(dpts), 0, ops._stream())
    def weights():
        lib.call("lush_mlp_bwd_weights", 0, H, H, R, S, C.byref(st), lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), C.byref(gs), 0, ops._stream())
    weights(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)(); L.lush_debug_prof_dw(out, 1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); weights(); e.record(); torch.cuda.synchronize()
    L.lush_debug_prof_dw(out, 1); v = list(out); ms = a.elapsed_time(e)
    jobs = max(v[8], 1)
    tick = ms * 1e3 / max(v[0], 1)      # us per s_memtime count, if workgroup 0 ran the whole launch
    print(f"R={R} S={S}: {ms:.3f} ms, kernel {v[0]} counts ({tick*1e3:.2f} ns each), {jobs} jobs; per job in us: "
          f"ring zero + record {v[1]*tick/jobs:.1f}, first encoding chunk {v[2]*tick/jobs:.1f}, first tile ready {v[3]*tick/jobs:.1f}, "
          f"steady loop {v[4]*tick/jobs:.1f}, drain {v[5]*tick/jobs:.1f}, flush {v[6]*tick/jobs:.1f}; sum {sum(v[1:7])*tick/jobs:.1f} of {v[0]*tick/jobs:.1f}")
    sp = (C.c_ulonglong * 2048)(); L.lush_debug_prof_dw_span(sp); sp = list(sp)
    n = 256
    # (s_memtime does not compare between CUs: only a workgroup's own span means something; all 256 start with the launch)
    dur = sorted((sp[1024 + b] - sp[b]) * tick for b in range(n))
    q = lambda a, f: a[int(f * (len(a) - 1))]
    print(f"   workgroup spans in us: min {q(dur, 0):.0f} 10% {q(dur, .1):.0f} median {q(dur, .5):.0f} 90% {q(dur, .9):.0f} max {q(dur, 1):.0f}; mean {sum(dur)/n:.0f}")
    byx = [sum((sp[1024 + b] - sp[b]) * tick for b in range(x, n, 8)) / (n // 8) for x in range(8)]
    print("   mean span by workgroup % 8:", " ".join(f"{e:.0f}" for e in byx))
