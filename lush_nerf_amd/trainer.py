"""Training step around the HIP ray march: the build's counterpart of the reference's
train() inner loop (run_lushnerf.py:603-685) and of its nn.DataParallel wrapper (:348).

* loss = 0.5*MSE + 0.5*L1 on rgb_blur and rgb0_blur (:652-661), plus 1e-2 * the consistency term of
  the aligned-pixel branch for i > noisenerf_start_iter (:629-661)
* Adam(lr 5e-4), lr = lrate * 0.1**(global_step / (lrate_decay*1000)) computed AFTER the step it
  follows, i.e. step g runs with the rate of global_step g-1 (:368-371, :675-685, :788)
* data parallel: one process per GPU, each draws its own N_rand rays; ONE all-reduce
  (RCCL over xGMI) of a single flat fp32 gradient buffer per step, then Adam runs
  redundantly on every rank.  No other collective on the data path;
  parameters and Adam moments are broadcast from rank 0 once at construction / checkpoint load
  (nn.DataParallel replicates module 0 every step, run_lushnerf.py:348).

All parameters live in one flat fp32 buffer (the nn.Parameters are views), ordered in three
Adam segments that mirror which parameters the reference leaves with grad=None
(SURVEY.md section 3.2): [coarse+fine MLP] always stepped; [RBK + noise MLP] only once the
blur kernel is on; [mlp_noise_coarse.alpha_linear] never.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .model import NeRFAll


class FlatParams:
    """Re-homes a module's parameters (and their .grad) into flat buffers."""

    def __init__(self, groups: List[List[torch.nn.Parameter]]):
        dev = groups[0][0].device
        seen, uniq_groups = set(), []
        for g in groups:
            ug = []
            for p in g:
                if id(p) not in seen:
                    seen.add(id(p))
                    ug.append(p)
            uniq_groups.append(ug)
        total = sum(p.numel() for g in uniq_groups for p in g)
        self.param = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.segments = []
        off = 0
        for g in uniq_groups:
            start = off
            for p in g:
                n = p.numel()
                self.param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.param[off:off + n].view_as(p.data)
                p.grad = self.grad[off:off + n].view_as(p.data)
                off += n
            self.segments.append((start, off))
        self.numel = total


def adam_segments(model: NeRFAll) -> List[List[torch.nn.Parameter]]:
    """[always stepped], [stepped once the blur kernel is on], [never stepped]."""
    base = list(model.mlp_coarse.parameters()) + (list(model.mlp_fine.parameters()) if model.mlp_fine else [])
    dead = list(model.mlp_noise_coarse.alpha_linear.parameters())
    dead_ids = {id(p) for p in dead}
    late = [p for p in model.mlp_noise_coarse.parameters() if id(p) not in dead_ids]
    if model.blur_kernel_net is not None:
        late = list(model.blur_kernel_net.parameters()) + late
    return [base, late, dead]


class Trainer:
    def __init__(self, model: NeRFAll, H: int, W: int, focal: float, N_samples: int = 64, N_importance: int = 64,
                 lrate: float = 5e-4, lrate_decay: int = 250, perturb: float = 1., raw_noise_std: float = 1.,
                 kernel_start_iter: int = 0, allkernel_start_iter: int = 0, noisenerf_start_iter: int = 1 << 30,
                 chunk: int = 1024 * 32, distributed: bool = False, micro_batch: int = 0, white_bkgd: bool = False,
                 step_fn=None, overlap: str = "auto"):
        self.model = model
        self.H, self.W = H, W
        self.K = [[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]]
        self.kw = dict(perturb=perturb, N_importance=N_importance, N_samples=N_samples, use_viewdirs=True,
                       white_bkgd=white_bkgd, raw_noise_std=raw_noise_std, inference=False, near=0., far=1.)
        # render_kwargs_test of the reference (run_lushnerf.py:406-410): what the consistency branch renders with
        self.kw_test = dict(self.kw, perturb=False, raw_noise_std=0., inference=True)
        self.lrate, self.lrate_decay = lrate, lrate_decay
        self.kernel_start_iter, self.allkernel_start_iter = kernel_start_iter, allkernel_start_iter
        self.noisenerf_start_iter = noisenerf_start_iter
        self.chunk = chunk
        # micro_batch > 0: forward+backward run per slice of that many INPUT rays and gradients accumulate in
        # the flat buffer (the loss is a mean over rays, so this is exact up to summation order).  It bounds the
        # activation stash: BASELINE config 5 (16 384 rays, 128+128) would otherwise hold ~260 GB between
        # forward and backward.
        self.micro_batch = micro_batch
        self.distributed = distributed and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        model.rng_stream = dist.get_rank() if self.distributed else 0     # ranks draw different jitter / noise
        self.flat = FlatParams(adam_segments(model))
        n = self.flat.numel
        self.m = torch.zeros(n, dtype=torch.float32, device=self.flat.param.device)
        self.v = torch.zeros_like(self.m)
        self.steps = [0, 0, 0]            # per-segment Adam step counters (torch keeps one per parameter)
        self.global_step = 0
        self._lr_next = None              # rate restored from a checkpoint's optimizer state, used for ONE step
        # test seam: the CPU (gloo) tests replace the HIP forward+backward of one slice by injected gradients
        self._fwd_bwd = step_fn or self._hip_forward_backward
        self._adam = ops.adam_step
        self.allreduce_events = None      # set to a list to collect (start, end) HIP events of every step's all-reduce
        # lush_march_bwd's two-stream overlap only pays when the runtime gives the two streams different
        # hardware queues: "auto" times steps 2-3 with it and 4-5 without (one device sync each, once) and keeps the faster;
        # "on" / "off" fix it; calibrate_overlap() does the same on demand (bench.py, inside its warm-up)
        self._overlap_mode = overlap
        self._overlap_probe = [] if overlap == "auto" else None
        if overlap == "off":
            self._set_overlap(False)
        self._graph = None                # step_graph: (key, torch.cuda.CUDAGraph, static batch, static loss, draw calls per step, active)
        self._graph_eager_left = 2        # plain steps before the capture (every lazy initialisation behind the entry points has run)
        self.sync_replicas()

    # ------------------------------------------------------------------ data parallel plumbing
    def sync_replicas(self):
        """Every rank continues from rank 0's parameters and Adam state (what nn.DataParallel's per-step
        replicate gives the reference for free).  Called at construction and after a checkpoint load."""
        if not self.distributed:
            return
        for t in (self.flat.param, self.m, self.v):
            dist.broadcast(t, 0)
        meta = torch.tensor(self.steps + [self.global_step], dtype=torch.int64, device=self.flat.param.device)
        dist.broadcast(meta, 0)
        meta = [int(x) for x in meta.tolist()]
        self.steps, self.global_step = meta[:3], meta[3]

    def replica_checksum(self) -> float:
        """max over ranks of |sum(param) - rank 0's sum(param)| (debug aid; 0.0 when replicas agree)."""
        s = self.flat.param.double().sum().reshape(1)
        if not self.distributed:
            return 0.0
        ref = s.clone()
        dist.broadcast(ref, 0)
        d = (s - ref).abs()
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        return float(d.item())

    def lr(self) -> float:
        """Rate the NEXT optimizer step runs with.  The reference sets new_lrate from global_step after
        optimizer.step() and before global_step += 1 (run_lushnerf.py:675-685, 788): step g (0-based) uses
        lrate * 0.1**((g-1)/decay), and the very first step the constructor's lrate."""
        g = max(self.global_step - 1, 0)
        return self.lrate * (0.1 ** (g / (self.lrate_decay * 1000)))

    # ------------------------------------------------------------------ one slice on the GPU
    def _hip_forward_backward(self, batch, a, b, i, draws, frac, force_naive):
        M = 1 if force_naive else self.model.mlp_rbk.num_motion + 1
        d = None if draws is None else {k: v[a * M:b * M] for k, v in draws.items()}
        rays = batch["rays"][a:b] if "rays" in batch else \
            ops.gen_rays(batch["c2w"], batch["view"][a:b], batch["px"][a:b], batch["py"][a:b], self.K)
        idx = batch["images_idx"][a:b] if "images_idx" in batch else batch["view"][a:b].reshape(-1, 1)
        out = self.model(self.H, self.W, self.K, chunk=self.chunk, rays=rays, rays_info={"images_idx": idx},
                         retraw=True, force_naive=force_naive, allkernel=i < self.allkernel_start_iter,
                         kernel_pixel=batch["fq_mask"][a:b], draws=d, **self.kw)
        part, ga, gb = ops.train_loss_grads(out[0], out[1], batch["target"][a:b], frac)
        torch.autograd.backward([out[0], out[1]], [ga, gb])
        return part

    def _consistency(self, consist, weight):
        """The aligned-pixel branch (run_lushnerf.py:629-650): loss_rgb and its backward; returns loss_rgb."""
        rgb_align, cert = self.model(self.H, self.W, self.K, self.chunk, poses=consist["poses"],
                                     render_kwargs=dict(self.kw_test), render_factor=0,
                                     rays_info=consist.get("images_idx"), consist_loss=True,
                                     Align_matrix=consist["Align_matrix"], Align_mask=consist["Align_mask"],
                                     anchor_pose=consist.get("anchor_pose"), samples=consist.get("samples"))
        loss_rgb = ops.ConsistLoss.apply(rgb_align, cert, 0.8)
        if weight != 0.0:
            (loss_rgb * weight).backward()
        return loss_rgb.detach()

    def step(self, batch: Dict[str, torch.Tensor], i: int, draws=None, consist: Optional[dict] = None):
        """One optimisation step on a batch {rays [N,3,2] (or c2w/view/px/py for device-side ray generation),
        images_idx [N,1], target [N,3], fq_mask [N]}."""
        self.model.train()
        if self._overlap_probe is not None:
            self._probe_overlap("begin")
        self.flat.grad.zero_()
        force_naive = i < self.kernel_start_iter
        N = batch["target"].shape[0]
        mb = self.micro_batch if 0 < self.micro_batch < N else N
        loss = None
        hooks = self.model.hooks
        sink_before, hooks.sink = hooks.sink, True
        try:      # dW kernels add straight into the flat gradient (p.grad are views of it)
            for a in range(0, N, mb):
                b = min(a + mb, N)
                part = self._fwd_bwd(batch, a, b, i, draws, (b - a) / N, force_naive)
                loss = part if loss is None else loss + part
            if consist is not None and i >= self.noisenerf_start_iter:
                # computed from i >= noisenerf_start_iter, added to the loss only for i > (run_lushnerf.py:629, 658-659)
                w = 1e-2 if i > self.noisenerf_start_iter else 0.0
                loss = loss + w * self._consistency(consist, w)
        finally:
            hooks.sink = sink_before
        if self.distributed:
            ev = None
            if self.allreduce_events is not None and self.flat.grad.is_cuda:      # bench.py: HIP events around the one collective
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(self.flat.grad)          # RCCL sum over xGMI; the 1/world mean is folded into Adam
            if ev is not None:
                ev[1].record()
                self.allreduce_events.append(ev)
        lr = self.lr() if self._lr_next is None else self._lr_next
        self._lr_next = None
        active = [True, not force_naive, False]
        for s, (a, b) in enumerate(self.flat.segments):
            if active[s] and b > a:
                self.steps[s] += 1
                self._adam(self.flat.param[a:b], self.flat.grad[a:b], self.m[a:b], self.v[a:b], lr, self.steps[s],
                           grad_scale=1.0 / self.world)
        self.global_step += 1
        if self._overlap_probe is not None:
            self._probe_overlap("end")
        return loss

    # ------------------------------------------------------------------ the step as one HIP graph
    def _step_body(self, batch, i, force_naive, state):
        """What step() enqueues, with rate / Adam step counts / draw counter taken from the device step state."""
        self.flat.grad.zero_()
        N = batch["target"].shape[0]
        mb = self.micro_batch if 0 < self.micro_batch < N else N
        loss = None
        hooks = self.model.hooks
        sink_before, hooks.sink = hooks.sink, True
        hooks.state, hooks.draw_delta = state, 0
        try:
            for a in range(0, N, mb):
                b = min(a + mb, N)
                part = self._fwd_bwd(batch, a, b, i, None, (b - a) / N, force_naive)
                loss = part if loss is None else loss + part
        finally:
            hooks.sink, hooks.state = sink_before, None
        calls = hooks.draw_delta
        active = [True, not force_naive, False]
        for s, (a, b) in enumerate(self.flat.segments):
            if active[s] and b > a:
                ops.adam_step_state(self.flat.param[a:b], self.flat.grad[a:b], self.m[a:b], self.v[a:b], state, s,
                                    grad_scale=1.0 / self.world)
        mask = sum(1 << s for s, (a, b) in enumerate(self.flat.segments) if active[s] and b > a)
        ops.lib.call("lush_step_state_advance", ops.lib.ptr(state), int(calls), int(mask), float(self.lrate),
                     float(self.lrate_decay * 1000), 0.9, 0.999, ops._stream())
        return loss, calls, mask

    def step_graph(self, batch: Dict[str, torch.Tensor], i: int):
        """step() with the whole step -- ray generation, both marches, loss, backward, Adam -- captured once in a HIP graph
        and replayed: one graph launch instead of ~50 kernel launches from Python (the launch-bound configurations: BASELINE
        config 1 spends its step in the host's launch path).  What the host passes per step as kernel arguments -- the
        learning rate, Adam's step counts, the Philox draw counter -- lives in the device step state (lush_step_state_*), which
        the graph's last node advances, so replays continue exactly where eager steps would: same draws, same rates.
        The first calls run step() itself (they are real steps); the batch is copied into the graph's static tensors before
        every replay; the returned loss
        is the graph's static tensor, overwritten by the next replay.  Single process only: a captured RCCL all-reduce has not
        been validated (world size 1 skips the all-reduce, which is the identity there); explicit draws, the consistency
        branch and a rate restored from a checkpoint fall back to step()."""
        force_naive = i < self.kernel_start_iter
        allk = i < self.allkernel_start_iter
        if self.world > 1:
            raise NotImplementedError("Trainer.step_graph: a captured RCCL all-reduce has not been validated; use step()")
        if self._lr_next is not None or i >= self.noisenerf_start_iter or self._graph_eager_left > 0 or not self.flat.param.is_cuda:
            self._graph_eager_left = max(self._graph_eager_left - 1, 0)
            return self.step(batch, i)
        key = (force_naive, allk, tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in batch.items())))
        hooks = self.model.hooks
        if self._graph is None or self._graph[0] != key:
            self.model.train()
            dev = self.flat.param.device
            state = torch.zeros(ops.lib.load().lush_step_state_bytes(), dtype=torch.uint8, device=dev)
            steps = (ops.C.c_int * 3)(*self.steps)
            ops.lib.call("lush_step_state_init", ops.lib.ptr(state), ops.C.c_ulonglong(hooks.draw_offset), int(self.global_step), steps,
                         float(self.lrate), float(self.lrate_decay * 1000), 0.9, 0.999, ops._stream())
            static = {k: v.clone() for k, v in batch.items()}
            graph = torch.cuda.CUDAGraph()
            distributed, self.distributed = self.distributed, False
            try:
                with torch.cuda.graph(graph):
                    loss, calls, mask = self._step_body(static, i, force_naive, state)
            finally:
                self.distributed = distributed
            self._graph = (key, graph, static, loss, calls, mask, state)
        key, graph, static, loss, calls, mask, state = self._graph
        for k, v in batch.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        for s in range(3):
            if (mask >> s) & 1:
                self.steps[s] += 1
        self.global_step += 1
        hooks.draw_offset += calls
        return loss

    def _set_overlap(self, on: bool):
        base = self.model.precision
        bit = 0 if on else ops.lib.VARIANT_NO_OVERLAP
        self.model.precision = ops.Precision(base.fwd, base.bwd, (base.variant & ~ops.lib.VARIANT_NO_OVERLAP) | bit)

    def _probe_overlap(self, phase: str):
        """step()'s passive calibration: called at the head ('begin') and the tail ('end') of the first six steps."""
        import time
        probe = self._overlap_probe
        dev = self.flat.param.device
        if probe is None or dev.type != "cuda" or self.model.precision.bwd != ops.PLANES_F16 or self.model.mlp_fine is None \
                or (self.model.precision.variant & ops.lib.VARIANT_NO_OVERLAP and not probe):
            self._overlap_probe = None
            return
        k = len(probe) // 2                      # index of the step being probed
        if phase == "begin":
            if k == 4:
                self._set_overlap(False)
            if k >= 2:
                torch.cuda.synchronize(dev)
            probe.append(time.perf_counter())
        else:
            if len(probe) // 2 >= 2:
                torch.cuda.synchronize(dev)
            probe.append(time.perf_counter())
            if len(probe) == 12:                 # six steps seen: 2-3 with the overlap, 4-5 without
                dt = [probe[2 * j + 1] - probe[2 * j] for j in range(6)]
                t = torch.tensor([dt[2] + dt[3], dt[4] + dt[5]], dtype=torch.float64, device=dev)
                if self.distributed:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                # (kept unless clearly slower: synchronised single steps scatter by +-1 %, the failure this guards against -- both
                # kernels in one hardware queue -- costs +12 %)
                self._set_overlap(bool(t[0] <= 1.03 * t[1]))
                self.overlap_probe_ms = [float(x) * 500.0 for x in t.tolist()]      # ms per step: with, without
                self._overlap_probe = None

    def calibrate_overlap(self, batches, i0: int = 0, steps: int = 2) -> dict:
        """lush_march_bwd runs the fine pass's weight gradients on a second stream beside the coarse pass's chain (two kernels
        that share the chip only if the runtime gives the two streams different hardware queues -- otherwise they run one after
        the other on part of the chip each, which is slower than not overlapping at all).  This times `steps` real steps each
        way on THIS process's streams, keeps the faster setting in the model's precision (all ranks take the slowest rank's
        view) and returns the two timings in ms per step.  The steps are ordinary optimisation steps."""
        import time
        dev = self.flat.param.device
        if not dev.type == "cuda":
            return {}
        self._overlap_probe = None            # (explicit calibration replaces the passive one of step())
        base = self.model.precision
        out = {}
        k = i0
        for name, bit in (("overlap", 0), ("one_after_the_other", ops.lib.VARIANT_NO_OVERLAP)):
            self.model.precision = ops.Precision(base.fwd, base.bwd, (base.variant & ~ops.lib.VARIANT_NO_OVERLAP) | bit)
            self.step(batches[k % len(batches)], k)       # (one untimed step: lazy initialisation of the second stream)
            k += 1
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(batches[k % len(batches)], k)
                k += 1
            torch.cuda.synchronize(dev)
            t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=dev)
            if self.distributed:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out[name] = float(t.item())
        keep = 0 if out["overlap"] <= 1.03 * out["one_after_the_other"] else ops.lib.VARIANT_NO_OVERLAP      # (see _probe_overlap)
        self.model.precision = ops.Precision(base.fwd, base.bwd, (base.variant & ~ops.lib.VARIANT_NO_OVERLAP) | keep)
        out["chosen"] = "overlap" if keep == 0 else "one_after_the_other"
        out["steps_taken"] = k - i0
        return out

    def faults(self) -> int:
        """Numerical-fault word of the model's render calls since the last read (one device sync; the reference
        prints after every chunk, models/lushnerf.py:474-478, 578-582)."""
        return self.model.read_faults()
