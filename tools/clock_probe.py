#!/usr/bin/env python3
"""Developer tool (needs a GPU and a -DLUSH_CLOCK build named by LUSH_SO): the clock each MLP kernel group actually runs at.

  python tools/build_variant.py --out build/clock.so --flags=-DLUSH_CLOCK
  LUSH_SO=build/clock.so python tools/clock_probe.py            > profiles/r05_clock.txt

MI355X_MICROARCH.md, "DVFS give-back" item 6: the in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped once
around the kernel after >= 2 s of back-to-back launches on random data, median over workgroups; board power and sclk are not the
test.  The stamps live in an array of their own (csrc/lush_common.h LUSH_CLOCK_*); the product build has none.
Per group (fine-pass shape, 20 480 rays x 128 samples = 2.62 M points, mode (h,h)): launches for SECONDS seconds, then the stamps of the
last launch.  Also the whole backward pair (chain then weights, alternating) as the training step runs them."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lush_nerf_amd import lib, ops, synth
if not os.environ.get("LUSH_SO"):
    raise SystemExit("clock_probe.py: name a -DLUSH_CLOCK build with LUSH_SO (tools/build_variant.py)")
lib.use_library(os.environ["LUSH_SO"])
from oracle import lush_oracle as O      # (developer tool: ray packing only)

dev = torch.device("cuda:0")
R, S = int(os.environ.get("R", 20480)), int(os.environ.get("S", 128))
SECONDS = float(os.environ.get("SECONDS", 2.5))
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
b = synth.ray_batch(R, 1)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
draw = torch.randn(R * S, 4, device=dev) * 1e-3
H = ops.PLANES_F16
so = C.CDLL(lib.SO_PATH)
L = lib.load()
pk = ops.mlp_pack(0, H, tens)
raw, stash = ops.mlp_forward(0, H, tens, pk, batch, z, True, H)
dstash = torch.empty(L.lush_mlp_dstash_bytes(0, H, R * S), dtype=torch.uint8, device=dev)
grads = [torch.zeros_like(t) for t in tens]
dpts = torch.empty(R * S, 8, device=dev)
st, gs = lib.mlp_struct(tens, 8), lib.mlp_struct(grads, 8)


def fwd():
    ops.mlp_forward(0, H, tens, pk, batch, z, True, H)


def fwd_nostash():
    ops.mlp_forward(0, H, tens, pk, batch, z, False, 0)


def chain():
    lib.call("lush_mlp_bwd_chain", 0, H, H, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pk), C.byref(st), lib.ptr(draw), lib.ptr(stash),
             lib.ptr(dstash), lib.ptr(dpts), 0, ops._stream())


def weights():
    lib.call("lush_mlp_bwd_weights", 0, H, H, R, S, C.byref(st), lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), C.byref(gs), 0, ops._stream())


def stamps(fn_name):
    out = (C.c_ulonglong * 4096)()
    rc = getattr(so, fn_name)(out)
    if rc != 0:
        raise SystemExit(f"{fn_name} failed")
    a = np.frombuffer(out, dtype=np.uint64).reshape(1024, 4).astype(np.float64)
    if os.environ.get("DUMP_WG_US") and "dw" in fn_name:      # per-workgroup durations in launch order (x + gridDim.x * y): job balance
        print(json.dumps({"dw_workgroup_us": [round(float(x), 1) for x in (a[:, 3] - a[:, 1]) / 100.0 if x > 0]}), flush=True)
    a = a[(a[:, 2] > a[:, 0]) & (a[:, 3] > a[:, 1])]
    mhz = (a[:, 2] - a[:, 0]) / (a[:, 3] - a[:, 1]) * 100.0
    us = (a[:, 3] - a[:, 1]) / 100.0
    return mhz, us


def run(label, fns, readers):
    for f in fns:
        f()
    torch.cuda.synchronize()
    for r in readers:
        stamps(r)                                     # (clears the arrays)
    t0 = time.perf_counter()
    n = 0
    ev = None
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(8):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in fns]
            for f, (a, e) in zip(fns, ev):
                a.record(); f(); e.record()
            n += 1
        torch.cuda.synchronize()
    ms = [a.elapsed_time(e) for a, e in ev]
    res = {"group": label, "launch_sets": n, "seconds": round(time.perf_counter() - t0, 2), "last_launch_ms": [round(x, 3) for x in ms]}
    for r in readers:
        mhz, us = stamps(r)
        res[r.replace("lush_debug_clock_", "")] = {"workgroups": int(mhz.size), "clock_MHz_median": round(float(np.median(mhz)), 1),
                                                   "clock_MHz_min": round(float(mhz.min()), 1), "clock_MHz_max": round(float(mhz.max()), 1),
                                                   "workgroup_us_median": round(float(np.median(us)), 1)}
    print(json.dumps(res), flush=True)


print(json.dumps({"device": torch.cuda.get_device_name(0), "R": R, "S": S, "points": R * S, "mode": "h,h",
                  "method": "d(s_memtime)/d(s_memrealtime) x 100 MHz per workgroup, last launch after SECONDS of back-to-back launches",
                  "SECONDS": SECONDS}), flush=True)
if os.environ.get("ONLY_DW"):
    run("weight gradients", [weights], ["lush_debug_clock_dw"])
    sys.exit(0)
run("forward, stash on (training)", [fwd], ["lush_debug_clock_fwd"])
run("forward, no stash (eval)", [fwd_nostash], ["lush_debug_clock_fwd"])
run("dX chain", [chain], ["lush_debug_clock_chain"])
run("weight gradients", [weights], ["lush_debug_clock_dw"])
run("backward pair as in a step (chain, weights alternating)", [chain, weights], ["lush_debug_clock_chain", "lush_debug_clock_dw"])
run("fwd, chain, weights alternating (a step's MLP launches)", [fwd, chain, weights], ["lush_debug_clock_fwd", "lush_debug_clock_chain", "lush_debug_clock_dw"])
