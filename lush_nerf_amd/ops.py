"""torch.autograd.Function wrappers over the HIP kernels (C ABI in include/lush_march.h).

torch is used for device memory, streams and autograd bookkeeping only; every
arithmetic step on the ray-march path is a hand-written gfx950 kernel.  Nothing
here falls back to torch ops or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import lib

NET_NERF, NET_NOISE = 0, 1
PLANES_F16 = 17          # plane code of the C ABI: 1..3 = bf16 planes, 17 = ONE fp16 plane (forward only)


def nplanes(code: int) -> int:
    return 1 if code == PLANES_F16 else code


def stash_code(pf: int, pb: int) -> int:
    """What the forward keeps for the backward: an fp16 forward stashes its single fp16 plane, otherwise the
    first min(pf, planes of the backward) bf16 planes (the dW GEMM converts between bf16 and fp16 in registers when the
    stash and the gradient chain differ)."""
    return PLANES_F16 if pf == PLANES_F16 else min(pf, nplanes(pb))


def parse_planes(text: str):
    """'2,1' / 'h,1' / '2,h' / 'h,h' -> (fwd code, bwd code); h = one fp16 plane (backward: loss-scaled)."""
    f, b = (PLANES_F16 if x.strip() in ("h", "17") else int(x) for x in text.split(","))
    return f, b
_NL = {NET_NERF: 8, NET_NOISE: 4}
N_MLP_TENSORS = {NET_NERF: 24, NET_NOISE: 16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    if not t.is_cuda:
        raise RuntimeError("lush_nerf_amd ops need CUDA/HIP tensors (no CPU path)")
    return t


@dataclass
class Precision:
    """16-bit planes per MFMA operand: forward / backward.  fwd = 2 (two bf16 planes, outputs within
    5e-7 of fp32) or PLANES_F16 (one fp16 plane, outputs within ~3e-5; the tiny noise MLP then stays on
    two bf16 planes) satisfy the 1e-4 bound; (1, 1) plain bf16 does not; (3, 3) is ~fp32.  The backward
    is bf16 (1 plane = plain bf16, 2 = fp32-equivalent) or PLANES_F16: ONE fp16 plane under a per-launch
    power-of-two loss scale chosen on the device from max|d_raw| -- 11-bit operands at the cost of the 8-bit ones."""
    fwd: int = 2
    bwd: int = 2
    variant: int = 0      # lib.VARIANT_* bits: an older kernel for the same work (A/B timing, agreement tests); 0 = the product's choice

    def noise(self) -> "Precision":
        return Precision(2, self.bwd, self.variant) if self.fwd == PLANES_F16 else self


# ----------------------------------------------------------------------------- MLP plumbing
def mlp_pack(net: int, planes: int, tensors: Sequence[torch.Tensor], variant: int = -1) -> torch.Tensor:
    """variant >= 0: only the copies the kernels of that variant read (lush_mlp_pack_for); default: every copy."""
    nbytes = lib.load().lush_mlp_packed_bytes(net, planes)
    out = torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device)
    st = lib.mlp_struct(tensors, _NL[net])
    lib.call("lush_mlp_pack_for", net, planes, C.byref(st), lib.ptr(out), int(variant), _stream())
    return out


class KernelTimer:
    """HIP-event timing of the three MLP kernel groups on the launch stream (bench.py)."""

    def __init__(self):
        self.spans = []      # (group, points, start_event, end_event)

    def span(self, group, points):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.append((group, points, a, b))
        return a, b

    def summary(self):
        out = {}
        for group, pts, a, b in self.spans:
            d = out.setdefault(group, {"launches": 0, "ms": 0.0, "points": 0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["points"] += pts
        return out


@dataclass
class Hooks:
    """Per-model switches of the ops (carried by NeRFAll / Trainer and handed to every op: nothing here is process-global,
    the reference calls its model from DataParallel worker threads).
      timer: bench.py -- HIP-event timing of the MLP kernel groups; the march then runs kernel group by kernel group
             through the piecewise entry points instead of the one-call lush_march_fwd / lush_march_bwd;
      keep:  tests -- a dict that receives the activation stashes of the last forward ("stash_c", "stash_f",
             "stash_noise" + the point counts), so a test can read the ReLU decisions the GPU took;
      sink:  Trainer.step -- parameter gradients are accumulated by the weight-gradient kernels' own atomics straight
             into each parameter's existing .grad buffer (the trainer's flat gradient) and autograd receives None for
             them: no per-tensor temporaries, zero-fills or `grad += tmp` kernels on the step;
      packed: Trainer.step -- the networks' MFMA fragments, packed once per step by one launch (PackPlan) and valid until the
             optimiser moves the parameters: the marches and the noise MLP of the step take them instead of re-packing per call;
      live_acc: bench.py -- an int64 [4] device tensor; the backward of a live-point march adds its passes' {live, all} point counts
             (one small add per march, no synchronisation): the share of points whose backward ran;
      draw_offset: Philox offset of the next lush_draws call (march_draws);
      state: Trainer.step_graph -- the device step state (include/lush_march.h lush_step_state_*): draws then take the
             state's counter plus `draw_delta`, the number of the call inside the step, so a step captured in a HIP graph
             draws fresh numbers on every replay (the same numbers the eager step would have drawn)."""
    timer: Optional[KernelTimer] = None
    keep: Optional[dict] = None
    live_acc: Optional[torch.Tensor] = None      # int64 [4]: every live-point march adds {live, all} points of its fine and coarse pass (bench.py)
    sink: bool = False
    packed: Optional[dict] = None      # Trainer.step: {(data_ptr of a net's first weight, plane code): packed fragments} of THIS step
    draw_offset: int = 0
    state: Optional[torch.Tensor] = None
    draw_delta: int = 0


class PackPlan:
    """Every network of a training step re-packed by ONE launch (include/lush_march.h lush_pack_plan_*): entries =
    [(net, plane code, parameter tensors, variant)].  Parameter and destination addresses are baked into the device-resident
    plan at construction (a synchronising set-up call); run() enqueues the one kernel; buffers maps
    (data_ptr of the net's first weight, plane code) -> packed fragments, what ops.Hooks.packed holds during a step."""

    def __init__(self, entries):
        L = lib.load()
        n = len(entries)
        dev = entries[0][2][0].device
        self.buffers: Dict[tuple, torch.Tensor] = {}
        self._keep = []
        jobs = (lib.PackJobC * n)()
        for i, (net, planes, tensors, variant) in enumerate(entries):
            buf = torch.empty(L.lush_mlp_packed_bytes(net, planes), dtype=torch.uint8, device=dev)
            st = lib.mlp_struct(tensors, _NL[net])
            self._keep.append(st)
            jobs[i].net, jobs[i].planes, jobs[i].variant = int(net), int(planes), int(variant)
            jobs[i].prm = C.pointer(st)
            jobs[i].packed = buf.data_ptr()
            self.buffers[(tensors[0].data_ptr(), int(planes))] = buf
        nbytes = L.lush_pack_plan_bytes(n)
        if nbytes == 0:
            raise RuntimeError("lush_pack_plan_bytes: 1 .. 8 jobs")
        self.plan = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        blocks = C.c_int(0)
        lib.call("lush_pack_plan_build", jobs, n, lib.ptr(self.plan), nbytes, C.byref(blocks))
        self.blocks = int(blocks.value)
        self.signature = tuple((int(net), int(planes), int(variant), tuple(t.data_ptr() for t in tensors)) for net, planes, tensors, variant in entries)

    def run(self, zero: Optional[torch.Tensor] = None):
        """Enqueue the one launch; `zero` (a contiguous fp32 tensor, e.g. the trainer's flat gradient) is cleared by the same kernel."""
        if zero is not None and (zero.dtype != torch.float32 or not zero.is_contiguous()):
            raise ValueError("PackPlan.run: the buffer to clear must be contiguous fp32")
        lib.call("lush_pack_plan_run", lib.ptr(self.plan), self.blocks, lib.ptr(zero) if zero is not None else None,
                 zero.numel() if zero is not None else 0, _stream())


def _packed_for(hooks, tensors, planes):
    """The step's packed fragments of (network, plane code) if the trainer packed them (ops.Hooks.packed), else None."""
    if hooks is None or hooks.packed is None:
        return None
    return hooks.packed.get((tensors[0].data_ptr(), int(planes)))


def mlp_forward(net: int, planes: int, tensors, packed, rays, z, want_stash: bool, stash_planes: int = 0, variant: int = 0,
                timer: Optional[KernelTimer] = None, group: str = "mlp_fwd"):
    """stash_planes: planes kept for the backward (default: all `planes`)."""
    R, S = z.shape
    sp = nplanes(stash_planes or planes) if want_stash else 0   # 0 = inference: only the gamma-row workspace
    raw = torch.empty(R * S, 4, dtype=torch.float32, device=rays.device)
    stash = torch.empty(lib.load().lush_mlp_stash_bytes(net, planes, sp, R * S), dtype=torch.uint8,
                        device=rays.device)
    st = lib.mlp_struct(tensors, _NL[net])
    ev = timer.span(group if net == NET_NERF else "noise_fwd", R * S) if timer is not None else None
    if ev:
        ev[0].record()
    lib.call("lush_mlp_fwd", net, planes, sp, lib.ptr(rays), lib.ptr(z), R, S, lib.ptr(packed), C.byref(st),
             lib.ptr(raw), lib.ptr(stash), int(variant), _stream())
    if ev:
        ev[1].record()
    return raw, (stash if want_stash else None)


def live_backward(precision: "Precision", hooks: Optional["Hooks"]) -> bool:
    """The march of this mode runs its backward on the live points only (include/lush_march.h "Live points"; the same rule as
    csrc/lush_march_abi.hip live_mode): one or two planes each way (the three-plane reference mode keeps the backward over all
    the points), the product's kernels, and no test hook that wants every point's stash."""
    older = lib.VARIANT_FWD_HALF | lib.VARIANT_FWD_512 | lib.VARIANT_BWD_HALF | lib.VARIANT_BWD_512 | lib.VARIANT_PE_ROWS | \
        lib.VARIANT_HEAD_KERNEL | lib.VARIANT_DW_SPLIT | lib.VARIANT_DENSE_BWD
    chain = (1, 2, PLANES_F16)
    return precision.fwd in chain and precision.bwd in chain and not (int(precision.variant) & older) and \
        (hooks is None or hooks.keep is None)


def live_compact(draw: torch.Tensor, R: int, S: int):
    """lush_live_compact: d_raw [R*S, 4] -> (live_idx int32 [R*S], draw_c [R*S, 4], ray_start int32 [R+1], cnt int32 [2] = {live, R*S})."""
    dev = draw.device
    P = R * S
    lidx = torch.empty(P, dtype=torch.int32, device=dev)
    draw_c = torch.empty(P, 4, dtype=torch.float32, device=dev)
    ray_start = torch.empty(R + 1, dtype=torch.int32, device=dev)
    cnt = torch.empty(2, dtype=torch.int32, device=dev)
    aux = torch.empty(max(int(lib.load().lush_live_aux_bytes(P)), 4), dtype=torch.uint8, device=dev)
    lib.call("lush_live_compact", lib.ptr(draw), R, S, lib.ptr(lidx), lib.ptr(draw_c), lib.ptr(ray_start), lib.ptr(cnt), lib.ptr(aux), _stream())
    return lidx, draw_c, ray_start, cnt


def mlp_forward_live(net: int, planes: int, tensors, packed, rays, z, lidx, cnt, stash_planes: int = 0, variant: int = 0,
                     timer: Optional[KernelTimer] = None):
    """The forward once more, WITH the stash, on the live points of a pass (lush_mlp_fwd_live): returns the stash.  With a timer the
    point count is read back (a synchronisation: the bench's kernel-group pass only)."""
    R, S = z.shape
    sp = nplanes(stash_planes or planes)
    stash = torch.empty(lib.load().lush_mlp_stash_bytes(net, planes, sp, R * S), dtype=torch.uint8, device=rays.device)
    st = lib.mlp_struct(tensors, _NL[net])
    ev = timer.span("mlp_fwd" if net == NET_NERF else "noise_fwd", int(cnt[0].item())) if timer is not None else None
    if ev:
        ev[0].record()
    lib.call("lush_mlp_fwd_live", net, planes, sp, lib.ptr(rays), lib.ptr(z), R, S, lib.ptr(packed), C.byref(st), lib.ptr(stash),
             lib.ptr(lidx), lib.ptr(cnt), int(variant), _stream())
    if ev:
        ev[1].record()
    return stash


def grad_sink(tensors, hooks: Optional[Hooks]):
    """The .grad buffers of `tensors` if accumulation into them is enabled (hooks.sink) and every one is usable, else None."""
    if hooks is None or not hooks.sink:
        return None
    gs = [getattr(t, "grad", None) for t in tensors]
    for g, t in zip(gs, tensors):
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != t.shape or g.device != t.device:
            return None
    return gs


def mlp_backward(net: int, planes_f: int, planes_b: int, tensors, packed_b, rays, z, draw, stash, sink=None, variant: int = 0,
                 timer: Optional[KernelTimer] = None, live=None):
    """Returns (list of parameter grads in `tensors` order, dpts [P][8]).  With `sink` (a list of fp32
    buffers, one per tensor) the gradients are ADDED to those buffers and the returned list holds None.
    live = (live_idx, cnt) of ops.live_compact: `draw` is the gathered d_raw, `stash` the live points' (mlp_forward_live), dpts
    comes back in list order (lush_mlp_bwd_chain_live / _weights_live)."""
    R, S = z.shape
    dev = rays.device
    dstash = torch.empty(lib.load().lush_mlp_dstash_bytes(net, planes_b, R * S), dtype=torch.uint8, device=dev)
    if sink is not None:
        grads = list(sink)
    else:
        flat = torch.zeros(sum(t.numel() for t in tensors), dtype=torch.float32, device=dev)   # one zero-fill for all of them
        grads, o = [], 0
        for t in tensors:
            grads.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
    dpts = torch.empty(R * S, 8, dtype=torch.float32, device=dev)
    st, gs = lib.mlp_struct(tensors, _NL[net]), lib.mlp_struct(grads, _NL[net])
    if live is not None:
        lidx, cnt = live
        npts = int(cnt[0].item()) if timer is not None else 0
        ev = timer.span("mlp_bwd_chain", npts) if timer is not None else None
        if ev:
            ev[0].record()
        lib.call("lush_mlp_bwd_chain_live", net, planes_f, planes_b, lib.ptr(rays), lib.ptr(z), R, S, lib.ptr(packed_b), C.byref(st), lib.ptr(draw),
                 lib.ptr(stash), lib.ptr(dstash), lib.ptr(dpts), lib.ptr(lidx), lib.ptr(cnt), int(variant), _stream())
        if ev:
            ev[1].record()
        ev = timer.span("mlp_bwd_weights", npts) if timer is not None else None
        if ev:
            ev[0].record()
        lib.call("lush_mlp_bwd_weights_live", net, planes_f, planes_b, R, S, C.byref(st), lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash),
                 C.byref(gs), lib.ptr(cnt), int(variant), _stream())
        if ev:
            ev[1].record()
        return ([None] * len(tensors) if sink is not None else grads), dpts
    timed = timer is not None and net == NET_NERF
    if not timed:      # chain + weight gradients as ONE call (the loss-scale launch then also zeroes the weight gradients' scratch)
        lib.call("lush_mlp_bwd", net, planes_f, planes_b, lib.ptr(rays), lib.ptr(z), R, S, lib.ptr(packed_b), C.byref(st), lib.ptr(draw),
                 lib.ptr(stash), lib.ptr(dstash), C.byref(gs), lib.ptr(dpts), int(variant), _stream())
        return ([None] * len(tensors) if sink is not None else grads), dpts
    ev = timer.span("mlp_bwd_chain", R * S) if timed else None
    if ev:
        ev[0].record()
    lib.call("lush_mlp_bwd_chain", net, planes_f, planes_b, lib.ptr(rays), lib.ptr(z), R, S, lib.ptr(packed_b),
             C.byref(st), lib.ptr(draw), lib.ptr(stash), lib.ptr(dstash), lib.ptr(dpts), int(variant), _stream())
    if ev:
        ev[1].record()
    ev = timer.span("mlp_bwd_weights", R * S) if timed else None
    if ev:
        ev[0].record()
    lib.call("lush_mlp_bwd_weights", net, planes_f, planes_b, R, S, C.byref(st), lib.ptr(draw), lib.ptr(stash),
             lib.ptr(dstash), C.byref(gs), int(variant), _stream())
    if ev:
        ev[1].record()
    return ([None] * len(tensors) if sink is not None else grads), dpts


# ----------------------------------------------------------------------------- ray prologue
class PackRays(torch.autograd.Function):
    """rays [...,3,2] -> ray batch [R,11] = [o, d, near, far, viewdir]: the head of
    render_infer/render_train_scene/render_train_noise (models/lushnerf.py:706-729)
    with ndc_rays (utils/run_lushnerf_helpers.py:542-562)."""

    @staticmethod
    def forward(ctx, rays, H, W, focal, ndc, near, far):
        rays = _f32(rays).reshape(-1, 3, 2)
        N = rays.shape[0]
        cx = float(torch.tensor(-1. / (W / (2. * focal)), dtype=torch.float32))
        cy = float(torch.tensor(-1. / (H / (2. * focal)), dtype=torch.float32))
        batch = torch.empty(N, 11, dtype=torch.float32, device=rays.device)
        lib.call("lush_pack_rays_fwd", lib.ptr(rays), N, int(bool(ndc)), cx, cy, float(near), float(far),
                 lib.ptr(batch), _stream())
        ctx.save_for_backward(rays)
        ctx.cfg = (int(bool(ndc)), cx, cy)
        return batch

    @staticmethod
    def backward(ctx, g):
        (rays,) = ctx.saved_tensors
        ndc, cx, cy = ctx.cfg
        g = _f32(g)
        d = torch.empty_like(rays)
        lib.call("lush_pack_rays_bwd", lib.ptr(rays), rays.shape[0], ndc, cx, cy, lib.ptr(g), lib.ptr(d), _stream())
        return d, None, None, None, None, None, None


# ----------------------------------------------------------------------------- the march
@dataclass
class MarchCfg:
    N_samples: int
    N_importance: int = 0
    perturb: float = 0.
    raw_noise_std: float = 0.
    white_bkgd: bool = False
    lindisp: bool = False
    near_mask: float = -1.0          # eval only: render_rmnearplane/128, <0 = off
    precision: Precision = None
    has_fine: bool = True
    want_grad: bool = True           # set by the caller from torch.is_grad_enabled() (it is off inside forward)
    flags: Optional[torch.Tensor] = None   # int32 [1] numerical-fault word (include/lush_march.h LUSH_FAULT_*), or None
    hooks: Optional[Hooks] = None

    def __post_init__(self):
        if self.precision is None:
            self.precision = Precision()
        if self.hooks is None:
            self.hooks = Hooks()


def zgrid(batch, S, lindisp, t_rand):
    R = batch.shape[0]
    z = torch.empty(R, S, dtype=torch.float32, device=batch.device)
    lib.call("lush_zgrid", lib.ptr(batch), R, S, int(lindisp), lib.ptr(t_rand), lib.ptr(z), _stream())
    return z


def composite_fwd(raw, z, batch, noise, cfg: MarchCfg, flag_shift: int = 0):
    R, S = z.shape
    dev = z.device
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    acc = torch.empty(R, dtype=torch.float32, device=dev)
    weights = torch.empty(R, S, dtype=torch.float32, device=dev)
    density = torch.empty(R, S - 1, dtype=torch.float32, device=dev)
    lib.call("lush_composite_fwd", lib.ptr(raw), lib.ptr(z), lib.ptr(batch), R, S, lib.ptr(noise),
             float(cfg.raw_noise_std), float(cfg.near_mask), int(cfg.white_bkgd), lib.ptr(rgb), lib.ptr(depth),
             lib.ptr(acc), lib.ptr(weights), lib.ptr(density), lib.ptr(cfg.flags), int(flag_shift), _stream())
    return rgb, depth, acc, weights, density


def composite_bwd(raw, z, batch, noise, cfg: MarchCfg, g_rgb, g_depth, g_acc, drays):
    R, S = z.shape
    draw = torch.empty(R * S, 4, dtype=torch.float32, device=z.device)
    lib.call("lush_composite_bwd", lib.ptr(raw), lib.ptr(z), lib.ptr(batch), R, S, lib.ptr(noise),
             float(cfg.raw_noise_std), float(cfg.near_mask), int(cfg.white_bkgd), lib.ptr(g_rgb), lib.ptr(g_depth),
             lib.ptr(g_acc), lib.ptr(draw), lib.ptr(drays), None, None, 0, 0, _stream())
    return draw


def sample_merge(z, weights, Ni, u, flags=None):
    R, S = z.shape
    dev = z.device
    z_out = torch.empty(R, S + Ni, dtype=torch.float32, device=dev)
    z_samples = torch.empty(R, Ni, dtype=torch.float32, device=dev)
    z_std = torch.empty(R, dtype=torch.float32, device=dev)
    lib.call("lush_sample_merge", lib.ptr(z), lib.ptr(weights), R, S, Ni, lib.ptr(u), lib.ptr(z_out),
             lib.ptr(z_samples), lib.ptr(z_std), lib.ptr(flags), _stream())
    return z_out, z_samples, z_std


def _opt(g):
    return None if g is None else _f32(g)


def _flat_grads(tensors, dev):
    """One zero-filled flat fp32 buffer and its per-tensor views."""
    flat = torch.zeros(sum(t.numel() for t in tensors), dtype=torch.float32, device=dev)
    views, o = [], 0
    for t in tensors:
        views.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return views


class March(torch.autograd.Function):
    """NeRFAll.render_rays_nonoise (models/lushnerf.py:481-583) as one differentiable op:
    z grid + jitter -> coarse MLP -> compositing -> sample_pdf + merge -> fine MLP ->
    compositing.  Differentiable w.r.t. the ray batch (columns 0..5, 8..10) and every MLP
    parameter; z_samples are detached exactly as in the reference (:546).

    Forward and backward are ONE C-ABI call each (lush_march_fwd / lush_march_bwd, include/lush_march.h) working out of
    one workspace tensor; with cfg.hooks.timer set (bench.py's kernel-group timing) the same kernels run through the
    piecewise entry points, bracketed by HIP events.

    Outputs: rgb, depth, acc, density, raw, weights, z_vals [, rgb0, depth0, acc0, density0, z_std].
    density / raw / weights / z_vals / z_std are returned non-differentiable (raw, weights, z_vals are views of the workspace).
    """

    @staticmethod
    def forward(ctx, batch, cfg: MarchCfg, draws: Dict[str, torch.Tensor], n_coarse: int, *params):
        ctx.set_materialize_grads(False)     # outputs nothing depends on arrive as None, not as zero tensors: a pass whose
        batch = _f32(batch)                  # outputs are all unused (the coarse net of the consistency branch) is skipped
        coarse = [_f32(p) for p in params[:n_coarse]]
        fine = [_f32(p) for p in params[n_coarse:]]
        same = not fine
        if same:
            fine = coarse
        need_grad = cfg.want_grad and any(ctx.needs_input_grad)
        d = {}
        if cfg.perturb > 0:
            d["t_rand"] = _opt(draws.get("t_rand"))
            d["u"] = _opt(draws.get("u")) if cfg.N_importance > 0 else None
        if cfg.raw_noise_std > 0:
            d["noise_c"] = _opt(draws.get("noise_c"))
            d["noise_f"] = _opt(draws.get("noise_f")) if cfg.N_importance > 0 else None
        ctx.cfg, ctx.n_coarse, ctx.n_params, ctx.same, ctx.need_grad = cfg, n_coarse, len(params), same, need_grad
        ctx.n_c = len(coarse)
        fwd = March._forward_piecewise if cfg.hooks.timer is not None else March._forward_fused
        outs, saved = fwd(ctx, batch, cfg, d, coarse, fine, same, need_grad)
        ctx.draw_keys = [k for k in ("t_rand", "noise_c", "u", "noise_f") if d.get(k) is not None]
        ctx.n_saved = len(saved)
        ctx.save_for_backward(batch, *saved, *[d[k] for k in ctx.draw_keys], *coarse, *([] if same else fine))
        nd = [outs[3], outs[4], outs[5], outs[6]] + ([outs[10], outs[11]] if cfg.N_importance > 0 else [])
        ctx.mark_non_differentiable(*nd)
        return tuple(outs)

    # ------------------------------------------------------------------ one call per direction
    @staticmethod
    def _packed_of(cfg: MarchCfg, same: bool, need_grad: bool, coarse, fine):
        """The step's packed fragments (ops.Hooks.packed, set by the trainer for the duration of a step) this march reads: looked
        up ONCE, at the forward, and kept by the autograd node -- the backward passes the same buffers whatever hooks.packed holds
        by then (a backward outside Trainer.step's try block, a retained graph): lush_march_fwd skips the workspace copy of
        fragments it is handed, so a backward that looked again and found nothing would chain over a never-written copy."""
        if cfg.hooks is None or cfg.hooks.packed is None:
            return None
        pf, pb = cfg.precision.fwd, cfg.precision.bwd
        pk = {"coarse": _packed_for(cfg.hooks, coarse, pf), "fine": None if same else _packed_for(cfg.hooks, fine, pf)}
        if need_grad and pb != pf:
            pk["bwd_coarse"] = _packed_for(cfg.hooks, coarse, pb)
            pk["bwd_fine"] = None if same else _packed_for(cfg.hooks, fine, pb)
        return pk

    @staticmethod
    def _c_cfg(cfg: MarchCfg, R: int, same: bool, need_grad: bool, packed=None):
        pf, pb = cfg.precision.fwd, cfg.precision.bwd
        variant = int(cfg.precision.variant)
        if cfg.hooks is not None and cfg.hooks.keep is not None:
            variant |= lib.VARIANT_DENSE_BWD      # tests read the ReLU decisions of EVERY point from the stash: the forward must keep it
        c = lib.MarchCfgC(R, cfg.N_samples, cfg.N_importance, float(cfg.perturb), float(cfg.raw_noise_std), int(cfg.white_bkgd),
                          int(cfg.lindisp), float(cfg.near_mask), pf, pb if need_grad else 0, variant, int(same))
        if packed is not None:      # packed once per step by the trainer (the tensors stay alive in `packed`, held by the caller)
            p = lambda k: None if packed.get(k) is None else packed[k].data_ptr()
            c.packed_coarse, c.packed_fine = p("coarse"), p("fine")
            if need_grad and pb != pf:
                c.packed_bwd_coarse, c.packed_bwd_fine = p("bwd_coarse"), p("bwd_fine")
        return c

    @staticmethod
    def _view(ws, c, which, shape):
        off, nbytes = C.c_size_t(), C.c_size_t()
        lib.call("lush_march_view", C.byref(c), which, C.byref(off), C.byref(nbytes))
        return ws[off.value:off.value + nbytes.value].view(torch.float32).view(*shape)

    @staticmethod
    def _forward_fused(ctx, batch, cfg, d, coarse, fine, same, need_grad):
        R, S, Ni, dev = batch.shape[0], cfg.N_samples, cfg.N_importance, batch.device
        Sl = S + Ni
        ctx.packed = March._packed_of(cfg, same, need_grad, coarse, fine)
        c = March._c_cfg(cfg, R, same, need_grad, ctx.packed)
        ctx.c_variant = int(c.variant)          # (the backward must lay the workspace out as the forward did)
        nbytes = lib.load().lush_march_workspace_bytes(C.byref(c))
        if nbytes == 0:
            raise RuntimeError("lush_march_workspace_bytes: bad configuration")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        rgb, depth, acc, density = f(R, 3), f(R), f(R), f(R, Sl - 1)
        o = lib.MarchOut(rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), density.data_ptr())
        if Ni > 0:
            rgb0, depth0, acc0, density0, z_std = f(R, 3), f(R), f(R), f(R, S - 1), f(R)
            o.rgb0, o.depth0, o.acc0, o.density0, o.z_std = (t.data_ptr() for t in (rgb0, depth0, acc0, density0, z_std))
        dr = lib.MarchDraws(*(None if d.get(k) is None else d[k].data_ptr() for k in ("t_rand", "noise_c", "u", "noise_f")))
        stc, stf = lib.mlp_struct(coarse, _NL[NET_NERF]), lib.mlp_struct(fine, _NL[NET_NERF])
        lib.call("lush_march_fwd", C.byref(c), lib.ptr(batch), C.byref(stc), C.byref(stf), C.byref(dr), C.byref(o), lib.ptr(ws),
                 lib.ptr(cfg.flags), _stream())
        outs = [rgb, depth, acc, density, March._view(ws, c, lib.VIEW_RAW, (R, Sl, 4)), March._view(ws, c, lib.VIEW_WEIGHTS, (R, Sl)),
                March._view(ws, c, lib.VIEW_Z, (R, Sl))]
        if Ni > 0:
            outs += [rgb0, depth0, acc0, density0, z_std]
        if cfg.hooks.keep is not None and need_grad:
            def raw_view(which):
                off, nb = C.c_size_t(), C.c_size_t()
                lib.call("lush_march_view", C.byref(c), which, C.byref(off), C.byref(nb))
                return ws[off.value:off.value + nb.value]
            cfg.hooks.keep.update(stash_c=raw_view(lib.VIEW_STASH_COARSE), stash_f=raw_view(lib.VIEW_STASH_FINE) if Ni > 0 else None,
                                  P_c=R * S, P_f=R * Sl if Ni > 0 else 0, batch=batch)
        ctx.fused = True
        return outs, [ws]

    @staticmethod
    def _backward_fused(ctx, g, batch, saved, d, coarse, fine):
        cfg = ctx.cfg
        (ws,) = saved
        R, fine_on = batch.shape[0], cfg.N_importance > 0
        c = March._c_cfg(cfg, R, ctx.same, True, ctx.packed)      # (the forward's packed buffers, not a second look at the hooks)
        c.variant = ctx.c_variant
        gp = [_opt(g[0]), _opt(g[1]), _opt(g[2])] + ([_opt(g[7]), _opt(g[8]), _opt(g[9])] if fine_on else [None, None, None])
        go = lib.MarchGout(*(None if t is None else t.data_ptr() for t in gp))
        any_main = any(t is not None for t in gp[:3])
        any_c = any(t is not None for t in gp[3:]) if fine_on else any_main
        ran_f, ran_c = fine_on and any_main, any_c

        def grads_for(tensors, ran):
            if not ran:
                return None, [None] * len(tensors)
            sink = grad_sink(tensors, cfg.hooks)
            if sink is not None:
                return sink, [None] * len(tensors)
            v = _flat_grads(tensors, batch.device)
            return v, v
        if ctx.same:
            buf_c, ret_c = grads_for(coarse, ran_f or ran_c)
            buf_f, ret_f = None, []
        else:
            buf_c, ret_c = grads_for(coarse, ran_c)
            buf_f, ret_f = grads_for(fine, ran_f)
        # (the first pass that runs writes d(ray batch) whole: no zero-fill launch; nothing runs when no output gradient arrived)
        drays = torch.empty_like(batch) if (ran_f or ran_c) else torch.zeros_like(batch)
        dr = lib.MarchDraws(*(None if d.get(k) is None else d[k].data_ptr() for k in ("t_rand", "noise_c", "u", "noise_f")))
        stc, stf = lib.mlp_struct(coarse, _NL[NET_NERF]), lib.mlp_struct(fine, _NL[NET_NERF])
        gc = lib.mlp_struct(buf_c, _NL[NET_NERF]) if buf_c is not None else None
        gf = lib.mlp_struct(buf_f, _NL[NET_NERF]) if buf_f is not None else None
        lib.call("lush_march_bwd", C.byref(c), lib.ptr(batch), C.byref(stc), C.byref(stf), C.byref(dr), C.byref(go), lib.ptr(ws),
                 None if gc is None else C.byref(gc), None if gf is None else C.byref(gf), lib.ptr(drays), _stream())
        if cfg.hooks.live_acc is not None and (ran_f or ran_c) and not (c.variant & lib.VARIANT_DENSE_BWD) and live_backward(cfg.precision, None):
            off, nb = C.c_size_t(), C.c_size_t()
            lib.call("lush_march_view", C.byref(c), lib.VIEW_LIVE_COUNTS, C.byref(off), C.byref(nb))
            cnt = ws[off.value:off.value + 16].view(torch.int32)
            if not (ran_f and ran_c) or not fine_on:      # a pass that did not run left its slot untouched
                cnt = cnt.clone()
                if not fine_on or not ran_c:
                    cnt[2:] = 0
                if fine_on and not ran_f:
                    cnt[:2] = 0
            cfg.hooks.live_acc += cnt
        return drays, ret_c, ret_f

    # ------------------------------------------------------------------ kernel group by kernel group (bench.py's timing pass)
    @staticmethod
    def _forward_piecewise(ctx, batch, cfg, d, coarse, fine, same, need_grad):
        pf, pb, var, tm = cfg.precision.fwd, cfg.precision.bwd, cfg.precision.variant, cfg.hooks.timer
        live = need_grad and live_backward(cfg.precision, cfg.hooks)      # the backward re-runs the forward on the live points: no stash here
        ctx.live = live
        stash_fwd = need_grad and not live
        grp = "mlp_fwd_all" if live else "mlp_fwd"
        zc = zgrid(batch, cfg.N_samples, cfg.lindisp, d.get("t_rand"))
        pk_c = mlp_pack(NET_NERF, pf, coarse, var)
        raw_c, stash_c = mlp_forward(NET_NERF, pf, coarse, pk_c, batch, zc, stash_fwd, stash_code(pf, pb), var, tm, grp)
        rgb, depth, acc, weights, density = composite_fwd(raw_c, zc, batch, d.get("noise_c"), cfg,
                                                          lib.FAULT_COARSE_SHIFT if cfg.N_importance > 0 else 0)
        outs = [rgb, depth, acc, density]
        z_last, raw_last, w_last = zc, raw_c, weights
        saved = [zc, raw_c, stash_c if stash_c is not None else torch.empty(0, device=batch.device)]
        if cfg.N_importance > 0:
            zf, _, z_std = sample_merge(zc, weights, cfg.N_importance, d.get("u"), cfg.flags)
            pk_f = pk_c if same else mlp_pack(NET_NERF, pf, fine, var)
            raw_f, stash_f = mlp_forward(NET_NERF, pf, fine, pk_f, batch, zf, stash_fwd, stash_code(pf, pb), var, tm, grp)
            rgb1, depth1, acc1, weights1, density1 = composite_fwd(raw_f, zf, batch, d.get("noise_f"), cfg)
            outs = [rgb1, depth1, acc1, density1]
            saved += [zf, raw_f, stash_f if stash_f is not None else torch.empty(0, device=batch.device)]
            z_last, raw_last, w_last = zf, raw_f, weights1
        R = batch.shape[0]
        outs += [raw_last.view(R, -1, 4), w_last, z_last]
        if cfg.N_importance > 0:
            outs += [rgb, depth, acc, density, z_std]
        if cfg.hooks.keep is not None:
            cfg.hooks.keep.update(stash_c=saved[2], stash_f=saved[5] if cfg.N_importance > 0 else None, P_c=zc.numel(),
                                  P_f=saved[3].numel() if cfg.N_importance > 0 else 0, batch=batch)
        ctx.fused = False
        return outs, saved

    @staticmethod
    def _backward_piecewise(ctx, g, batch, saved, d, coarse, fine):
        cfg = ctx.cfg
        pf, pb, tm = cfg.precision.fwd, cfg.precision.bwd, cfg.hooks.timer
        drays = torch.zeros_like(batch)
        fine_on = cfg.N_importance > 0
        g_main = (_opt(g[0]), _opt(g[1]), _opt(g[2]))
        g_c = (_opt(g[7]), _opt(g[8]), _opt(g[9])) if fine_on else g_main
        grads_c: List[Optional[torch.Tensor]] = [None] * len(coarse)
        grads_f: List[Optional[torch.Tensor]] = []

        def run(tensors, z, raw, noise, stash, gg):
            draw = composite_bwd(raw, z, batch, noise, cfg, gg[0], gg[1], gg[2], drays)
            pk = mlp_pack(NET_NERF, pb, tensors, cfg.precision.variant)
            if getattr(ctx, "live", False):      # (the one-call march does the same inside lush_march_bwd)
                lidx, draw_c, ray_start, cnt = live_compact(draw, z.shape[0], z.shape[1])
                pk_f = pk if pf == pb else mlp_pack(NET_NERF, pf, tensors, cfg.precision.variant)      # (the forward's own fragments)
                st_live = mlp_forward_live(NET_NERF, pf, tensors, pk_f, batch, z, lidx, cnt, stash_code(pf, pb), cfg.precision.variant, tm)
                gr, dpts = mlp_backward(NET_NERF, stash_code(pf, pb), pb, tensors, pk, batch, z, draw_c, st_live,
                                        sink=grad_sink(tensors, cfg.hooks), variant=cfg.precision.variant, timer=tm, live=(lidx, cnt))
                lib.call("lush_ray_grad_reduce_live", lib.ptr(dpts), lib.ptr(z), lib.ptr(lidx), lib.ptr(ray_start), z.shape[0], lib.ptr(drays), _stream())
                return gr
            gr, dpts = mlp_backward(NET_NERF, stash_code(pf, pb), pb, tensors, pk, batch, z, draw, stash,
                                    sink=grad_sink(tensors, cfg.hooks), variant=cfg.precision.variant, timer=tm)
            lib.call("lush_ray_grad_reduce", lib.ptr(dpts), lib.ptr(z), z.shape[0], z.shape[1], lib.ptr(drays), _stream())
            return gr

        if fine_on and any(x is not None for x in g_main):
            grads_f = run(fine, saved[3], saved[4], d.get("noise_f"), saved[5], g_main)
        if any(x is not None for x in g_c):
            grads_c = run(coarse, saved[0], saved[1], d.get("noise_c"), saved[2], g_c)
        if fine_on and ctx.same and grads_f:
            grads_c = [(a + b if b is not None else a) if a is not None else b for a, b in zip(grads_c, grads_f)]
            grads_f = []
        return drays, grads_c, (grads_f if not ctx.same else [])

    @staticmethod
    def backward(ctx, *g):
        t = ctx.saved_tensors
        batch, saved = t[0], list(t[1:1 + ctx.n_saved])
        o = 1 + ctx.n_saved
        d = {k: t[o + i] for i, k in enumerate(ctx.draw_keys)}
        o += len(ctx.draw_keys)
        coarse = list(t[o:o + ctx.n_c])
        fine = coarse if ctx.same else list(t[o + ctx.n_c:])
        bwd = March._backward_fused if ctx.fused else March._backward_piecewise
        drays, grads_c, grads_f = bwd(ctx, g, batch, saved, d, coarse, fine)
        n_fine = ctx.n_params - ctx.n_coarse
        out_f = list(grads_f) if grads_f else [None] * n_fine
        return (drays, None, None, None, *grads_c, *out_f[:n_fine])


class NoiseMlp(torch.autograd.Function):
    """NeRFAll.render_rays_noise + mlpforward_noise (models/lushnerf.py:268-293, 585-617):
    one NeRF_Noise evaluation per ray at sample `index` (16) of the un-jittered grid.  The
    ray batch is detached in the reference (:614), so only parameters receive gradients."""

    @staticmethod
    def forward(ctx, batch, N_samples, index, lindisp, precision: Precision, want_grad, hooks: Optional[Hooks], *params):
        batch = _f32(batch)
        tensors = [_f32(p) for p in params]
        R = batch.shape[0]
        z = torch.empty(R, 1, dtype=torch.float32, device=batch.device)
        lib.call("lush_zfixed", lib.ptr(batch), R, int(N_samples), int(index), int(lindisp), lib.ptr(z), _stream())
        pk = _packed_for(hooks, tensors, precision.fwd)
        if pk is None:
            pk = mlp_pack(NET_NOISE, precision.fwd, tensors)
        need = bool(want_grad) and any(ctx.needs_input_grad)
        raw, stash = mlp_forward(NET_NOISE, precision.fwd, tensors, pk, batch, z, need, stash_code(precision.fwd, precision.bwd),
                                 precision.variant, hooks.timer if hooks is not None else None)
        if hooks is not None and hooks.keep is not None:
            hooks.keep.update(stash_noise=stash, P_noise=R)
        ctx.save_for_backward(batch, z, *([stash] if stash is not None else []), *tensors)
        ctx.has_stash, ctx.precision, ctx.hooks = stash is not None, precision, hooks
        return raw[:, :3]           # a view of the [R,4] raw output (no copy launch; ops.BlurMix reads it with its row stride)

    @staticmethod
    def backward(ctx, g):
        pr = ctx.precision
        t = ctx.saved_tensors
        batch, z = t[0], t[1]
        stash = t[2] if ctx.has_stash else None
        tensors = list(t[3 if ctx.has_stash else 2:])
        base = g._base if g._is_view() else None
        if base is not None and getattr(base, "_lush_col3_zero", False) and base.dtype == torch.float32 \
                and tuple(base.shape) == (g.shape[0], 4) and base.is_contiguous() and g.data_ptr() == base.data_ptr() \
                and tuple(g.stride()) == (4, 1):
            draw = base             # ops.BlurMix.backward hands over columns 0..2 of a [R,4] buffer whose column 3 it zeroed -- and
                                    # says so on the buffer: any other [:, :3] view of a [R,4] tensor gets the zero-padded copy
        else:
            draw = torch.zeros(g.shape[0], 4, dtype=torch.float32, device=g.device)
            draw[:, :3] = g
        pk = _packed_for(ctx.hooks, tensors, pr.bwd)
        if pk is None:
            pk = mlp_pack(NET_NOISE, pr.bwd, tensors)
        grads, _ = mlp_backward(NET_NOISE, stash_code(pr.fwd, pr.bwd), pr.bwd, tensors, pk, batch, z, draw, stash,
                                sink=grad_sink(tensors, ctx.hooks), variant=pr.variant)
        o = 2 * _NL[NET_NOISE] + 4   # alpha_linear is dead in NeRF_Noise (helpers:496,505,512): grad None
        grads[o] = None
        grads[o + 1] = None
        return (None, None, None, None, None, None, None, *grads)


# ----------------------------------------------------------------------------- blur kernel
RBK_ACT = 512
RBK_RVW = 32
RBK_RVW_OFFSET = 480     # include/lush_march.h LUSH_RBK_RVW_OFFSET


class RbkWarp(torch.autograd.Function):
    """View_Embedding + Rigid_Blurring_Kernel.forward (models/lushnerf.py:27-35, 118-153):
    rays [N,3,2], images_idx [N] -> new_rays [N*(M+1),3,2], ccw [N,M+1].  The MLP depends on
    the image index only, so it runs once per image; the SE(3) warp runs per ray."""

    @staticmethod
    def forward(ctx, rays, idx, num_motion, window, mask, hooks: Optional[Hooks], *params):
        rays = _f32(rays).reshape(-1, 3, 2)
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        tensors = [_f32(p) for p in params]
        N, M = rays.shape[0], int(num_motion)
        num_img = tensors[0].shape[0]
        dev = rays.device
        acts = torch.empty(num_img, RBK_ACT, dtype=torch.float32, device=dev)
        st = lib.rbk_struct(tensors)
        lib.call("lush_rbk_mlp_fwd", C.byref(st), num_img, M, float(window), lib.ptr(acts), _stream())
        new_rays = torch.empty(N * (M + 1), 3, 2, dtype=torch.float32, device=dev)
        ccw = torch.empty(N, M + 1, dtype=torch.float32, device=dev)
        lib.call("lush_rbk_warp_fwd", lib.ptr(rays), lib.ptr(idx), N, M, lib.ptr(acts), lib.ptr(new_rays),
                 lib.ptr(ccw), _stream())
        if mask is not None:
            mask = mask.reshape(-1).to(torch.uint8).contiguous()
        ctx.save_for_backward(rays, idx, acts, *([mask] if mask is not None else []), *tensors)
        ctx.has_mask, ctx.hooks = mask is not None, hooks
        ctx.cfg = (N, M, num_img, float(window))
        return new_rays, ccw

    @staticmethod
    def backward(ctx, g_rays, g_ccw):
        N, M, num_img, window = ctx.cfg
        t = ctx.saved_tensors
        rays, idx, acts = t[0], t[1], t[2]
        mask = t[3] if ctx.has_mask else None
        tensors = list(t[4 if ctx.has_mask else 3:])
        dev = rays.device
        # gradients w.r.t. r, v, w accumulate in the zero tail of the activation rows (include/lush_march.h LUSH_RBK_RVW_OFFSET):
        # no buffer of their own, no zero-fill launch
        d_rvw = acts.view(-1)[RBK_RVW_OFFSET:]
        drays = torch.empty_like(rays) if ctx.needs_input_grad[0] else None
        lib.call("lush_rbk_warp_bwd", lib.ptr(rays), lib.ptr(idx), N, M, lib.ptr(acts),
                 lib.ptr(_opt(g_rays)), lib.ptr(_opt(g_ccw)), lib.ptr(mask), lib.ptr(d_rvw), RBK_ACT, lib.ptr(drays),
                 _stream())
        grads = _rbk_param_grads(tensors, ctx.hooks, num_img, M, window, acts, d_rvw)
        return (drays, None, None, None, None, None, *grads)


def _rbk_param_grads(tensors, hooks, num_img, M, window, acts, d_rvw):
    """lush_rbk_mlp_bwd into the trainer's flat gradient (hooks.sink: returns Nones) or into fresh tensors."""
    sink = grad_sink(tensors, hooks)
    grads = list(sink) if sink is not None else [torch.empty_like(x) for x in tensors]
    scratch = torch.empty(num_img, RBK_ACT, dtype=torch.float32, device=acts.device)
    st, gs = lib.rbk_struct(tensors), lib.rbk_struct(grads)
    lib.call("lush_rbk_mlp_bwd", C.byref(st), num_img, M, window, lib.ptr(acts), lib.ptr(d_rvw), RBK_ACT,
             C.byref(gs), lib.ptr(scratch), int(sink is not None), _stream())
    return [None] * len(grads) if sink is not None else grads


def _ndc_consts(H, W, focal):
    cx = float(torch.tensor(-1. / (W / (2. * focal)), dtype=torch.float32))
    cy = float(torch.tensor(-1. / (H / (2. * focal)), dtype=torch.float32))
    return cx, cy


class RbkWarpNdc(torch.autograd.Function):
    """RbkWarp followed by PackRays in one kernel per direction (SURVEY.md section 7.2 `rbk_warp_ndc`): View_Embedding +
    Rigid_Blurring_Kernel.forward (models/lushnerf.py:27-35, 118-153) and the head of render_train_scene / render_train_noise
    (:772-795, 827-850; ndc_rays helpers:542-562).  rays [N,3,2], images_idx -> ray batch [N*(M+1),11] of the warped rays,
    ccw [N,M+1], ray batch [N,11] of the input rays (what the noise branch marches; non-differentiable: the reference
    detaches it, :614).  The warped rays themselves never exist in memory."""

    @staticmethod
    def forward(ctx, rays, idx, num_motion, window, mask, hooks: Optional[Hooks], H, W, focal, ndc, near, far, *params):
        rays = _f32(rays).reshape(-1, 3, 2)
        idx = idx.reshape(-1)
        if idx.dtype != torch.int64 or not idx.is_contiguous():
            idx = idx.to(torch.int64).contiguous()
        tensors = [_f32(p) for p in params]
        N, M = rays.shape[0], int(num_motion)
        num_img = tensors[0].shape[0]
        dev = rays.device
        acts = torch.empty(num_img, RBK_ACT, dtype=torch.float32, device=dev)
        st = lib.rbk_struct(tensors)
        lib.call("lush_rbk_mlp_fwd", C.byref(st), num_img, M, float(window), lib.ptr(acts), _stream())
        cx, cy = _ndc_consts(H, W, focal)
        batch = torch.empty(N * (M + 1), 11, dtype=torch.float32, device=dev)
        ccw = torch.empty(N, M + 1, dtype=torch.float32, device=dev)
        batch0 = torch.empty(N, 11, dtype=torch.float32, device=dev)
        lib.call("lush_rbk_warp_ndc_fwd", lib.ptr(rays), lib.ptr(idx), N, M, lib.ptr(acts), int(bool(ndc)), cx, cy, float(near),
                 float(far), lib.ptr(batch), lib.ptr(ccw), lib.ptr(batch0), _stream())
        if mask is not None:
            mask = mask.reshape(-1)
            if mask.dtype != torch.uint8 or not mask.is_contiguous():
                mask = mask.to(torch.uint8).contiguous()
        ctx.save_for_backward(rays, idx, acts, *([mask] if mask is not None else []), *tensors)
        ctx.has_mask, ctx.hooks = mask is not None, hooks
        ctx.cfg = (N, M, num_img, float(window), int(bool(ndc)), cx, cy)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(batch0)
        return batch, ccw, batch0

    @staticmethod
    def backward(ctx, g_batch, g_ccw, _g_batch0):
        N, M, num_img, window, ndc, cx, cy = ctx.cfg
        t = ctx.saved_tensors
        rays, idx, acts = t[0], t[1], t[2]
        mask = t[3] if ctx.has_mask else None
        tensors = list(t[4 if ctx.has_mask else 3:])
        d_rvw = acts.view(-1)[RBK_RVW_OFFSET:]
        drays = torch.empty_like(rays) if ctx.needs_input_grad[0] else None
        lib.call("lush_rbk_warp_ndc_bwd", lib.ptr(rays), lib.ptr(idx), N, M, lib.ptr(acts), ndc, cx, cy, lib.ptr(_opt(g_batch)),
                 lib.ptr(_opt(g_ccw)), lib.ptr(mask), lib.ptr(d_rvw), RBK_ACT, lib.ptr(drays), num_img, _stream())
        grads = _rbk_param_grads(tensors, ctx.hooks, num_img, M, window, acts, d_rvw)
        return (drays, None, None, None, None, None, None, None, None, None, None, None, *grads)


class BlurMix(torch.autograd.Function):
    """The tail of NeRFAll.forward's training branch in one kernel per direction (SURVEY.md section 7.2 `blur_mix_tonemap`;
    models/lushnerf.py:644-654): rbk_weighted_sum of the fine and coarse colours (:100-116), rgb_noise = 0.1 sigmoid(noise_raw),
    tone mapping (helpers:164-174).  rgb, rgb0 [N*(M+1),3], ccw [N,M+1], noise_raw [N,3] (rows may be strided: the noise MLP's
    [N,4] raw output is read in place) -> (tm(rgb_pure + rgb_noise), tm(rgb0_pure + rgb_noise), rgb_noise, tm(rgb_pure),
    tm(rgb0_pure))."""

    @staticmethod
    def forward(ctx, rgb, rgb0, ccw, nraw, gamma):
        rgb, rgb0, ccw = _f32(rgb), _f32(rgb0), _f32(ccw)
        if nraw.dtype != torch.float32 or nraw.dim() != 2 or nraw.stride(1) != 1 or not nraw.is_cuda:
            nraw = _f32(nraw)
        N, M1 = ccw.shape
        out = torch.empty(5, N, 3, dtype=torch.float32, device=rgb.device)
        lib.call("lush_blur_mix_fwd", lib.ptr(rgb), lib.ptr(rgb0), lib.ptr(ccw), lib.ptr(nraw), int(nraw.stride(0)), N, M1,
                 int(bool(gamma)), *(lib.ptr(out[i]) for i in range(5)), _stream())
        ctx.save_for_backward(rgb, rgb0, ccw, nraw)
        ctx.gamma = int(bool(gamma))
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, *g):
        rgb, rgb0, ccw, nraw = ctx.saved_tensors
        N, M1 = ccw.shape
        d_rgb, d_rgb0 = torch.empty_like(rgb), torch.empty_like(rgb0)
        d_ccw = torch.empty_like(ccw)
        d_nraw4 = torch.empty(N, 4, dtype=torch.float32, device=rgb.device)      # column 3 = 0: the noise MLP's d_raw as it stands
        lib.call("lush_blur_mix_bwd", lib.ptr(rgb), lib.ptr(rgb0), lib.ptr(ccw), lib.ptr(nraw), int(nraw.stride(0)), N, M1, ctx.gamma,
                 *(lib.ptr(_opt(x)) for x in g), lib.ptr(d_rgb), lib.ptr(d_rgb0), lib.ptr(d_ccw), lib.ptr(d_nraw4), _stream())
        d_nraw4._lush_col3_zero = True      # (read by NoiseMlp.backward: the kernel wrote column 3 = 0)
        return d_rgb, d_rgb0, d_ccw, d_nraw4[:, :3], None


class WSum(torch.autograd.Function):
    """Rigid_Blurring_Kernel.rbk_weighted_sum for one tensor (models/lushnerf.py:100-116)."""

    @staticmethod
    def forward(ctx, x, ccw):
        x, ccw = _f32(x), _f32(ccw)
        N, M = ccw.shape
        shape = x.shape
        C_ = x.numel() // (N * M)
        y = torch.empty((N,) + tuple(shape[1:]), dtype=torch.float32, device=x.device)
        lib.call("lush_wsum_fwd", lib.ptr(x), lib.ptr(ccw), N, M, C_, lib.ptr(y), _stream())
        ctx.save_for_backward(x, ccw)
        ctx.dims = (N, M, C_)
        return y

    @staticmethod
    def backward(ctx, g):
        x, ccw = ctx.saved_tensors
        N, M, C_ = ctx.dims
        g = _f32(g)
        dx = torch.empty_like(x)
        dccw = torch.empty_like(ccw)
        lib.call("lush_wsum_bwd", lib.ptr(x), lib.ptr(ccw), N, M, C_, lib.ptr(g), lib.ptr(dx), lib.ptr(dccw), _stream())
        return dx, dccw


class ToneMap(torch.autograd.Function):
    """tonemapping(x [+ 0.1*sigmoid(noise_raw)]) for 'gamma' / 'none'
    (utils/run_lushnerf_helpers.py:164-174; models/lushnerf.py:649, 654)."""

    @staticmethod
    def forward(ctx, x, nraw, gamma):
        x = _f32(x)
        nraw = None if nraw is None else _f32(nraw)
        y = torch.empty_like(x)
        lib.call("lush_tonemap_fwd", lib.ptr(x), lib.ptr(nraw), x.numel() // 3, int(gamma), lib.ptr(y), _stream())
        ctx.save_for_backward(x, *([nraw] if nraw is not None else []))
        ctx.gamma = int(gamma)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        x = ctx.saved_tensors[0]
        nraw = ctx.saved_tensors[1] if len(ctx.saved_tensors) > 1 else None
        dx = torch.empty_like(x)
        dn = None if nraw is None else torch.empty_like(nraw)
        lib.call("lush_tonemap_bwd", lib.ptr(x), lib.ptr(nraw), x.numel() // 3, ctx.gamma, lib.ptr(g),
                 lib.ptr(dx), lib.ptr(dn), _stream())
        return dx, dn, None


class NoiseAct(torch.autograd.Function):
    """0.1 * sigmoid(x), models/lushnerf.py:649, 660."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        lib.call("lush_noise_act_fwd", lib.ptr(x), x.numel(), lib.ptr(y), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        lib.call("lush_noise_act_bwd", lib.ptr(x), x.numel(), lib.ptr(_f32(g)), lib.ptr(dx), _stream())
        return dx


def train_loss_grads(a, b, target, scale: float = 1.0, work: Optional[torch.Tensor] = None):
    """run_lushnerf.py:652-661 without an autograd node: (scale * loss, scale * d loss/d a, scale * d loss/d b) from one
    kernel; with `a is b` (no fine pass: rgb0 = rgb) the second gradient is None and the first is the sum of both terms'.  The trainer feeds the two gradients to torch.autograd.backward itself (no ones-fill, no grad * g kernels).
    work: 2 floats of scratch owned by the caller, zero before the first call (the kernel leaves them zero): the loss word is
    then written by the kernel instead of zero-filled here and accumulated."""
    same = a is b
    a, b, target = _f32(a.detach()), _f32(b.detach()), _f32(target)
    loss = torch.empty(1, dtype=torch.float32, device=a.device) if work is not None else torch.zeros(1, dtype=torch.float32, device=a.device)
    if same and a.data_ptr() == b.data_ptr():      # one tensor in both roles (no fine pass): ONE gradient, the sum of the two terms'
        ga = torch.empty_like(a)
        lib.call("lush_loss_fwd_bwd", lib.ptr(a), lib.ptr(a), lib.ptr(target), a.shape[0], float(scale), lib.ptr(loss),
                 lib.ptr(ga), None, lib.ptr(work), _stream())
        return loss[0], ga, None
    ga, gb = torch.empty_like(a), torch.empty_like(b)
    lib.call("lush_loss_fwd_bwd", lib.ptr(a), lib.ptr(b), lib.ptr(target), a.shape[0], float(scale), lib.ptr(loss),
             lib.ptr(ga), lib.ptr(gb), lib.ptr(work), _stream())
    return loss[0], ga, gb


class TrainLoss(torch.autograd.Function):
    """run_lushnerf.py:652-661: 0.5*MSE + 0.5*L1 on rgb_blur and on rgb0_blur."""

    @staticmethod
    def forward(ctx, a, b, target):
        a, b, target = _f32(a), _f32(b), _f32(target)
        loss = torch.zeros(1, dtype=torch.float32, device=a.device)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        lib.call("lush_loss_fwd_bwd", lib.ptr(a), lib.ptr(b), lib.ptr(target), a.shape[0], 1.0, lib.ptr(loss), lib.ptr(ga),
                 lib.ptr(gb), None, _stream())
        ctx.save_for_backward(ga, gb)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ga, gb = ctx.saved_tensors
        return ga * g, gb * g, None


def adam_step(param, grad, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """torch.optim.Adam semantics on flat fp32 buffers (run_lushnerf.py:368-371)."""
    lib.call("lush_adam", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), param.numel(), float(lr),
             float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream())


def adam_step_state(param, grad, m, v, state, segment, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """The same step with rate and bias corrections read from the device step state (Trainer.step_graph)."""
    lib.call("lush_adam_state", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), param.numel(), lib.ptr(state), int(segment),
             float(beta1), float(beta2), float(eps), float(grad_scale), _stream())


def adam_step_multi(param, grad, m, v, ends, mask, lr, steps, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """adam_step over the consecutive segments [0, ends[0]), [ends[0], ends[1]), [ends[1], ends[2]) of one flat buffer in one launch,
    segment s at its own step count steps[s]; bit s of `mask` clear = segment s is skipped."""
    st = (C.c_int * 3)(*[int(x) for x in steps])
    lib.call("lush_adam_multi", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), int(ends[0]), int(ends[1]), int(ends[2]), int(mask),
             float(lr), float(beta1), float(beta2), float(eps), st, float(grad_scale), _stream())


def adam_step_state_multi(param, grad, m, v, ends, state, mask, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """adam_step_state over the consecutive segments [0, ends[0]), [ends[0], ends[1]), [ends[1], ends[2]) of one flat buffer in one
    launch; bit s of `mask` clear = segment s is skipped."""
    lib.call("lush_adam_state_multi", lib.ptr(param), lib.ptr(grad), lib.ptr(m), lib.ptr(v), int(ends[0]), int(ends[1]), int(ends[2]),
             lib.ptr(state), int(mask), float(beta1), float(beta2), float(eps), float(grad_scale), _stream())


def gen_rays(c2w, view, px, py, K):
    """Device-side get_rays for (view, pixel) pairs: c2w [V,3,4], view/px/py [N] -> rays [N,3,2]
    (utils/run_lushnerf_helpers.py:517-539)."""
    c2w = _f32(c2w[:, :3, :4])
    view, px, py = (t.reshape(-1).to(torch.int64).contiguous() for t in (view, px, py))
    N = view.numel()
    rays = torch.empty(N, 3, 2, dtype=torch.float32, device=c2w.device)
    lib.call("lush_gen_rays", lib.ptr(c2w), lib.ptr(view), lib.ptr(px), lib.ptr(py), N, float(K[0][0]), float(K[1][1]),
             float(K[0][2]), float(K[1][2]), lib.ptr(rays), _stream())
    return rays


def gen_rays_image(c2w, H, W, K):
    """get_rays(H, W, K, c2w) of the eval path (utils/run_lushnerf_helpers.py:517-528; models/lushnerf.py:881) for
    every pixel of one pose: c2w [3(+),4] -> rays [H, W, 3, 2]."""
    c2w = _f32(c2w[:3, :4])
    rays = torch.empty(H, W, 3, 2, dtype=torch.float32, device=c2w.device)
    lib.call("lush_gen_rays_image", lib.ptr(c2w), int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]),
             float(K[1][2]), lib.ptr(rays), _stream())
    return rays


def align_rays(c2w, align, cert, samples, H, W, K):
    """Ray gather of Render_Aligned_Pixel (models/lushnerf.py:958-985): c2w [V,3(+),4], align [V,HW,4] =
    Align_matrix[anchor], cert [V,HW] = Align_mask[anchor] (bool / uint8 / float), samples [ns] int64
    -> rays [V*ns,3,2], certainty [V,ns] fp32."""
    c2w = _f32(c2w[:, :3, :4])
    align = _f32(align)
    V, HW = align.shape[0], align.shape[1]
    if align.shape[2] != 4 or c2w.shape[0] != V or tuple(cert.shape) != (V, HW):
        raise ValueError("align_rays: need align [V,HW,4], cert [V,HW], c2w [V,3,4]")
    is_u8 = cert.dtype in (torch.bool, torch.uint8)
    cert = cert.contiguous() if is_u8 else _f32(cert)
    if not cert.is_cuda:
        raise RuntimeError("lush_nerf_amd ops need CUDA/HIP tensors (no CPU path)")
    samples = samples.reshape(-1).to(device=c2w.device, dtype=torch.int64).contiguous()
    ns = samples.numel()
    rays = torch.empty(V * ns, 3, 2, dtype=torch.float32, device=c2w.device)
    cert_out = torch.empty(V, ns, dtype=torch.float32, device=c2w.device)
    lib.call("lush_align_rays", lib.ptr(c2w), lib.ptr(align), lib.ptr(cert), int(is_u8), lib.ptr(samples), V, ns, HW,
             int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), lib.ptr(rays),
             lib.ptr(cert_out), _stream())
    return rays, cert_out


class ConsistLoss(torch.autograd.Function):
    """loss_rgb of the consistency branch (run_lushnerf.py:644-650) with compute_mean_with_confidence
    (utils/run_lushnerf_helpers.py:665-688): rgb_align [V,ns,3], align_certainty [V,ns], threshold -> scalar."""

    @staticmethod
    def forward(ctx, rgb, cert, threshold):
        rgb, cert = _f32(rgb), _f32(cert)
        V, ns = cert.shape
        loss = torch.empty(1, dtype=torch.float32, device=rgb.device)
        grad = torch.empty_like(rgb)
        lib.call("lush_consist_loss_fwd_bwd", lib.ptr(rgb), lib.ptr(cert), V, ns, float(threshold), lib.ptr(loss),
                 lib.ptr(grad), _stream())
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def march_draws(R, N_samples, N_importance, perturb, raw_noise_std, device, seed=None, stream_id=0, hooks: Optional[Hooks] = None, offset=None):
    """The random draws of one march, in the reference's shapes (models/lushnerf.py:515, :322;
    utils/run_lushnerf_helpers.py:578), from ONE Philox launch (lush_draws) instead of four torch RNG kernels.
    seed defaults to torch's CUDA seed (so torch.manual_seed governs it); every call takes the next offset of `hooks`
    (the model's own counter; an explicit `offset` overrides it); stream_id separates data-parallel ranks that share a seed."""
    if seed is None:
        seed = torch.cuda.initial_seed()
    state = None
    if offset is None:
        if hooks is None:
            raise ValueError("march_draws: give the model's ops.Hooks (its draw counter) or an explicit offset")
        if hooks.state is not None:        # counter on the device: offset = its value at the head of the step + this call's number
            hooks.draw_delta += 1
            offset, state = hooks.draw_delta, hooks.state
        else:
            hooks.draw_offset += 1
            offset = hooks.draw_offset
    d = {}
    if perturb > 0:
        d["t_rand"] = torch.empty(R, N_samples, dtype=torch.float32, device=device)
    if raw_noise_std > 0:
        d["noise_c"] = torch.empty(R, N_samples - 1, dtype=torch.float32, device=device)
    if N_importance > 0:
        if perturb > 0:
            d["u"] = torch.empty(R, N_importance, dtype=torch.float32, device=device)
        if raw_noise_std > 0:
            d["noise_f"] = torch.empty(R, N_samples + N_importance - 1, dtype=torch.float32, device=device)
    if d:
        g = lambda k: (lib.ptr(d.get(k)), d[k].numel() if k in d else 0)
        if state is None:
            lib.call("lush_draws", C.c_ulonglong(int(seed) & (2 ** 64 - 1)),
                     C.c_ulonglong((int(stream_id) << 40) + int(offset)), *g("t_rand"), *g("noise_c"), *g("u"), *g("noise_f"),
                     _stream())
        else:
            lib.call("lush_draws_state", C.c_ulonglong(int(seed) & (2 ** 64 - 1)),
                     C.c_ulonglong((int(stream_id) << 40) + int(offset)), lib.ptr(state), *g("t_rand"), *g("noise_c"), *g("u"),
                     *g("noise_f"), _stream())
    return d
