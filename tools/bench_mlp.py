#!/usr/bin/env python3
"""Developer micro-benchmark of the MLP kernel groups (needs a GPU).  Not a test."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lush_nerf_amd import lib, ops, synth
if os.environ.get("LUSH_SO"):        # developer tool: a variant built by tools/build_variant.py
    lib.use_library(os.environ["LUSH_SO"])
from oracle import lush_oracle as O

dev = torch.device("cuda:0")
R = int(os.environ.get("R", 20480)); S = int(os.environ.get("S", 128))
modes = [ops.parse_planes(m) for m in os.environ.get("MODES", "2,2;1,1").split(";")]
what = os.environ.get("WHAT", "fwd,fwd_nostash,chain,weights").split(",")
reps = int(os.environ.get("REPS", 3))
variant = int(os.environ.get("VARIANT", 0))        # lib.VARIANT_* bits (A/B timing of kernel variants)
w = synth.all_weights(30, 0)
names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
tens = [torch.from_numpy(w[n]).to(dev) for n in names]
if os.environ.get("ZERO_W"):          # same instruction stream on all-zero weights and biases: what the clock does when the operands cost nothing
    tens = [torch.zeros_like(t) for t in tens]
b = synth.ray_batch(R, 1)
batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
z = torch.sort(torch.rand(R, S, device=dev), -1)[0]
draw = torch.randn(R * S, 4, device=dev) * 1e-3
MACS = 593408

def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return min(ts)

for nf, nb in modes:
    pk = ops.mlp_pack(0, nf, tens)
    pkb = pk if nb == nf else ops.mlp_pack(0, nb, tens)
    res = {}
    if "fwd" in what:
        res["fwd"] = timeit(lambda: ops.mlp_forward(0, nf, tens, pk, batch, z, True, ops.stash_code(nf, nb), variant))
    if "fwd_nostash" in what:
        res["fwd_nostash"] = timeit(lambda: ops.mlp_forward(0, nf, tens, pk, batch, z, False, 0, variant))
    raw, stash = ops.mlp_forward(0, nf, tens, pk, batch, z, True, ops.stash_code(nf, nb), variant)
    nf_s = ops.stash_code(nf, nb)
    import ctypes as C
    dstash = torch.empty(lib.load().lush_mlp_dstash_bytes(0, nb, R * S), dtype=torch.uint8, device=dev)
    grads = [torch.zeros_like(t) for t in tens]
    dpts = torch.empty(R * S, 8, device=dev)
    st, gs = lib.mlp_struct(tens, 8), lib.mlp_struct(grads, 8)
    def chain():
        lib.call("lush_mlp_bwd_chain", 0, nf_s, nb, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pkb), C.byref(st),
                 lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), lib.ptr(dpts), variant, ops._stream())
    def weights():
        lib.call("lush_mlp_bwd_weights", 0, nf_s, nb, R, S, C.byref(st), lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), C.byref(gs), variant, ops._stream())
    if "chain" in what:
        res["chain"] = timeit(chain)
    else:
        chain()
    if "weights" in what:
        res["weights"] = timeit(weights)
    fl = 2 * MACS * R * S / 1e9
    print(json.dumps({"planes": [nf, nb], "variant": variant, "R": R, "S": S, **{k: {"ms": round(v, 3), "alg_TF": round(fl / v, 1)} for k, v in res.items()}}), flush=True)
