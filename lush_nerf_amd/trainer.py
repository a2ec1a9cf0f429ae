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


class BlobBatch(dict):
    """A training batch whose per-step tensors are views of ONE contiguous device buffer, so that a captured step
    (Trainer.step_graph, bench.py's config-1 graph) refreshes its static inputs with one device copy instead of one per
    tensor (each ~5 us of GPU time: at config 1 six of them were 7 % of the step).  Keys in `shared` (the pose table) are
    referenced as they are and copied only when the caller hands over a different tensor."""

    def __init__(self, batch, shared=("c2w",), _blob=None):
        super().__init__()
        self.shared = tuple(k for k in shared if k in batch)
        per_step = [(k, v) for k, v in batch.items() if k not in self.shared]
        dev = per_step[0][1].device
        al = lambda n: (n + 15) // 16 * 16
        total = sum(al(v.numel() * v.element_size()) for _, v in per_step)
        self.blob = torch.empty(total, dtype=torch.uint8, device=dev) if _blob is None else _blob
        off = 0
        for k, v in per_step:
            nb = v.numel() * v.element_size()
            view = self.blob[off:off + nb].view(v.dtype).view(v.shape)
            if _blob is None:
                view.copy_(v)
            self[k] = view
            off += al(nb)
        for k in self.shared:
            self[k] = batch[k]

    def clone_static(self) -> "BlobBatch":
        return BlobBatch(self, self.shared, _blob=self.blob.clone())

    def load(self, other: "BlobBatch"):
        self.blob.copy_(other.blob, non_blocking=True)
        for k in self.shared:
            if self[k].data_ptr() != other[k].data_ptr():
                self[k].copy_(other[k], non_blocking=True)


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
                 step_fn=None):
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
        sg = self.flat.segments      # (three groups laid out one after the other: one Adam launch covers the active ones)
        self._segments_consecutive = len(sg) == 3 and sg[0][0] == 0 and sg[1][0] == sg[0][1] and sg[2][0] == sg[1][1]
        self.allreduce_events = None      # set to a list to collect (start, end) HIP events of every step's all-reduce
        # scratch of the loss kernel's reduction (zero once, left zero by every launch: no zero-fill per step)
        self._loss_work = torch.zeros(2, dtype=torch.float32, device=self.flat.param.device) if self.flat.param.is_cuda else None
        # live-point backward (include/lush_march.h "Live points"): the share of MLP evaluation points whose d_raw row is non-zero,
        # counted on the device by every march and followed on the host WITHOUT blocking (snapshots through pinned memory)
        self.live_policy = "auto"         # "auto": dense backward while the share stays above LIVE_MAX_SHARE; "live" / "dense": always
        self._live_cum = torch.zeros(4, dtype=torch.int64, device=self.flat.param.device) if self.flat.param.is_cuda else None
        self._live_ring = torch.zeros(8, 4, dtype=torch.int64).pin_memory() if self.flat.param.is_cuda else None
        self._live_slot = 0
        self._live_snaps = []             # [(event, ring row, step number, accumulator id)], oldest first
        self._live_prev = [0, 0, 0, 0]    # last snapshot read (cumulative counts)
        self._live_prev_acc = None        # ... and the accumulator it was read from (a caller may hand the march its own)
        self._live_step = 0               # steps whose backward choice has been taken
        self.live_share = None            # share of live points in the most recent step whose counts have arrived
        self._dense_now = False
        self._dense_steps = 0
        self._eager_seen = set()          # backward choices (dense: bool) that have run as an eager step (step_graph captures only those)
        self._pack_plan = None            # ops.PackPlan of the model's networks in the current precision mode
        self._pack_plans = {}             # ... by signature (precision mode, kernel variant, parameter addresses)
        self._graph = None                # step_graph: the captured step (graphs, static batch / loss, device step state + host mirror)
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
        self.invalidate_graph()

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

    # ------------------------------------------------------------------ small launches: 128-point tiles
    # A march whose largest MLP launch has at most this many points runs on the 128-point-tile kernels (two workgroups per CU:
    # lib.VARIANT_FWD_HALF | VARIANT_BWD_HALF) instead of the 256-point-tile ones: BASELINE config 1's 8 192 points are 32 of
    # the big tiles on a 256-CU chip, 64 of the small ones.  Measured (profiles/r05_c1_sweep.txt, one pass of R rays x 32
    # samples, forward / chain / weight gradients in us): R = 256: 72 / 72 / 71 -> 67 / 58 / 70; R = 1 024 (32 768 points):
    # 77 / 80 / 118 -> 74 / 69 / 115; R = 2 048: 92 / 96 / 189 -> 110 / 109 / 160 (the big tiles win from there).
    SMALL_LAUNCH_POINTS = 32768

    # The live-point march costs one forward over all the points more than the dense one and saves (1 - share) of the stash-
    # writing forward, the chain and the weight gradients: measured on BASELINE config 2 over 420 - 600 training steps
    # (profiles/r05_live_points.md), 4.4 .. 5.3 + 14.5 .. 17.4 x share ms against 16.1 .. 16.7 ms: break-even at a share of
    # 0.69 .. 0.76.  Above LIVE_MAX_SHARE the trainer runs the dense backward and looks again (one live step, which at such a
    # share costs what a dense one does) every LIVE_PROBE_EVERY steps; it returns to the live march below LIVE_MIN_SHARE.
    # (Round 5 switched at 0.65 / 0.60 and probed every 32 steps: inside the band where the live march is still the cheaper one,
    # and up to 32 steps late on a falling share.)
    LIVE_MAX_SHARE, LIVE_MIN_SHARE, LIVE_PROBE_EVERY = 0.70, 0.65, 8
    # The choice for step i is taken from the counts of step i - LIVE_LAG, whatever the host's and the device's relative timing:
    # the trainer WAITS for that step's snapshot (two steps behind the one being enqueued: it has all but always arrived, and the
    # host never gets more than LIVE_LAG steps ahead of the device), so a run with a fixed seed makes the same choices every time
    # (round 5 polled with event.query(): which steps ran dense depended on how far the host was ahead).
    LIVE_LAG = 2

    def _live_poll(self):
        """Consume the count snapshots of the steps up to LIVE_LAG behind the one that starts now (waits for them: see LIVE_LAG):
        updates live_share."""
        while self._live_snaps and self._live_snaps[0][2] <= self._live_step - self.LIVE_LAG:
            ev, row, _, acc_id = self._live_snaps.pop(0)
            ev.synchronize()
            cur = [int(x) for x in self._live_ring[row].tolist()]
            if acc_id != self._live_prev_acc:          # another accumulator (a caller's own, e.g. one per step): a series that starts at zero
                self._live_prev, self._live_prev_acc = [0, 0, 0, 0], acc_id
            d_live = (cur[0] - self._live_prev[0]) + (cur[2] - self._live_prev[2])
            d_all = (cur[1] - self._live_prev[1]) + (cur[3] - self._live_prev[3])
            if d_all > 0:
                self.live_share = d_live / d_all
            self._live_prev = cur

    def _live_snapshot(self, acc=None):
        """Enqueue a copy of the cumulative counts the step's marches added to (the trainer's accumulator, or the one a caller put
        into hooks.live_acc) into the next row of the pinned ring (no allocation, nothing blocks)."""
        acc = self._live_cum if acc is None else acc
        if acc is None or self._live_ring is None or len(self._live_snaps) >= self._live_ring.shape[0]:
            return                         # (cannot happen with LIVE_LAG < ring rows; a skipped step's delta folds into the next snapshot)
        row = self._live_slot
        self._live_slot = (row + 1) % self._live_ring.shape[0]
        self._live_ring[row].copy_(acc, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if self._live_prev_acc is None:
            self._live_prev_acc = acc.data_ptr()       # (the first series starts from zero counts)
        self._live_snaps.append((ev, row, self._live_step, acc.data_ptr()))

    def live_counts(self):
        """Cumulative {live, all} point counts of the fine and the coarse passes of every live-point march so far (a
        synchronisation; bench.py reads it around its timed region)."""
        return [0, 0, 0, 0] if self._live_cum is None else [int(x) for x in self._live_cum.tolist()]

    def live_policy_state(self) -> dict:
        """Where the policy stands (checkpoint.save_checkpoint keeps it; the advisor's note of round 5: none of it was part of any
        saved state).  The snapshots still in flight belong to steps the file does not contain and are dropped on load."""
        return {"policy": self.live_policy, "dense_now": bool(self._dense_now), "dense_steps": int(self._dense_steps),
                "live_share": None if self.live_share is None else float(self.live_share)}

    def load_live_policy_state(self, st: Optional[dict]):
        self._live_snaps, self._live_step = [], 0
        if self._live_cum is not None:
            self._live_prev, self._live_prev_acc = [int(x) for x in self._live_cum.tolist()], self._live_cum.data_ptr()
        if st is None:
            self._dense_now, self._dense_steps, self.live_share = False, 0, None
            return
        # (`policy` in the file is a record of how the saved run was set up; the resuming caller's own live_policy stays)
        self._dense_now, self._dense_steps, self.live_share = bool(st["dense_now"]), int(st["dense_steps"]), st["live_share"]

    def _dense_backward_now(self) -> bool:
        """The policy's choice for the step that starts now."""
        if self.live_policy == "dense":
            return True
        if self.live_policy == "live" or self.live_share is None:
            return False
        if self._dense_now:
            self._dense_steps += 1
            if self._dense_steps % self.LIVE_PROBE_EVERY == 0:
                return False                                   # one live step: look at the share again
            if self.live_share < self.LIVE_MIN_SHARE:
                self._dense_now = False
        elif self.live_share > self.LIVE_MAX_SHARE:
            self._dense_now, self._dense_steps = True, 0
        return self._dense_now

    def _largest_launch(self, n_rays: int, force_naive: bool, coarse_only: bool = False) -> int:
        M = 1 if (force_naive or coarse_only or self.model.blur_kernel_net is None) else self.model.mlp_rbk.num_motion + 1
        Ni = 0 if coarse_only else int(self.kw["N_importance"])
        return int(n_rays) * M * (int(self.kw["N_samples"]) + Ni)

    def _headline_default(self) -> bool:
        pr = self.model.precision
        return pr.variant == 0 and pr.fwd == ops.PLANES_F16 and pr.bwd == ops.PLANES_F16

    def _live_mode(self) -> bool:
        """The model's precision mode runs the live-point backward (and no explicit kernel variant is set)."""
        pr = self.model.precision
        return pr.variant == 0 and ops.live_backward(pr, None)

    def _backward_choice(self, n_rays: int, force_naive: bool) -> bool:
        """True: the step that starts now runs the dense backward.  Decided on the host from the counts that have arrived, OUTSIDE
        any graph capture (step_graph keeps one captured step per answer)."""
        if not self._live_mode() or self._live_cum is None:
            return False
        if self._headline_default() and self._largest_launch(n_rays, force_naive) <= self.SMALL_LAUNCH_POINTS:
            return False                                       # (the small-launch kernels: no live-point form)
        self._live_step += 1
        self._live_poll()
        return self._dense_backward_now()

    def _step_precision(self, n_rays: int, force_naive: bool, coarse_only: bool = False, dense: bool = False):
        """The precision mode this step's kernels run in: the model's, with the small-launch kernel variant when the step's
        largest MLP launch is small (headline mode only), or with the dense backward when _backward_choice said so (an explicit
        variant is left alone)."""
        pr = self.model.precision
        if self._headline_default() and self._largest_launch(n_rays, force_naive, coarse_only) <= self.SMALL_LAUNCH_POINTS:
            return ops.Precision(pr.fwd, pr.bwd, ops.lib.VARIANT_FWD_HALF | ops.lib.VARIANT_BWD_HALF)
        if dense and self._live_mode():
            return ops.Precision(pr.fwd, pr.bwd, ops.lib.VARIANT_DENSE_BWD)
        return pr

    class _PrecisionFor:
        """with self._PrecisionFor(model, precision): the model's ops run in `precision` (packing and every launch of the step
        must agree on the kernel variant: the stash formats differ)."""

        def __init__(self, model, precision):
            self.model, self.precision = model, precision

        def __enter__(self):
            self.saved, self.model.precision = self.model.precision, self.precision

        def __exit__(self, *a):
            self.model.precision = self.saved

    # ------------------------------------------------------------------ weights packed once per step
    def _pack_entries(self):
        pr = self.model.precision
        pn = pr.noise()
        ent = [(ops.NET_NERF, pr.fwd, self.model.mlp_coarse.tensors(), pr.variant)]
        if self.model.mlp_fine is not None:
            ent.append((ops.NET_NERF, pr.fwd, self.model.mlp_fine.tensors(), pr.variant))
        if pr.bwd != pr.fwd:
            ent.append((ops.NET_NERF, pr.bwd, self.model.mlp_coarse.tensors(), pr.variant))
            if self.model.mlp_fine is not None:
                ent.append((ops.NET_NERF, pr.bwd, self.model.mlp_fine.tensors(), pr.variant))
        ent.append((ops.NET_NOISE, pn.fwd, self.model.mlp_noise_coarse.tensors(), -1))
        if pn.bwd != pn.fwd:
            ent.append((ops.NET_NOISE, pn.bwd, self.model.mlp_noise_coarse.tensors(), -1))
        return ent

    def _pack_weights(self, zero_grad=False):
        """ONE launch re-packs every network's MFMA fragments for this step (ops.PackPlan; the plan is rebuilt when the precision
        mode or a parameter's address changes) and hands them to the step's ops through the model's hooks.  With zero_grad the
        same launch clears the flat gradient buffer (the step's zero_grad); without the HIP path it is cleared here."""
        if not self.flat.param.is_cuda or self._fwd_bwd != self._hip_forward_backward:
            if zero_grad:
                self.flat.grad.zero_()
            return
        ent = self._pack_entries()
        sig = tuple((int(n), int(p), int(v), tuple(t.data_ptr() for t in ts)) for n, p, ts, v in ent)
        if self._pack_plan is None or self._pack_plan.signature != sig:
            # (a plan per signature: a trainer whose steps alternate between two kernel variants does not rebuild -- a
            # synchronising set-up call -- every time)
            plan = self._pack_plans.get(sig)
            if plan is None:
                if len(self._pack_plans) >= 4:
                    self._pack_plans.clear()
                plan = self._pack_plans[sig] = ops.PackPlan(ent)
            self._pack_plan = plan
        self._pack_plan.run(self.flat.grad if zero_grad else None)
        self.model.hooks.packed = self._pack_plan.buffers

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
        part, ga, gb = ops.train_loss_grads(out[0], out[1], batch["target"][a:b], frac, work=self._loss_work)
        if gb is None:      # (one tensor in both roles)
            torch.autograd.backward([out[0]], [ga])
        else:
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

    def step(self, batch: Dict[str, torch.Tensor], i: int, draws=None, consist: Optional[dict] = None, _dense: Optional[bool] = None):
        """One optimisation step on a batch {rays [N,3,2] (or c2w/view/px/py for device-side ray generation),
        images_idx [N,1], target [N,3], fq_mask [N]}."""
        self.model.train()
        force_naive = i < self.kernel_start_iter
        N = batch["target"].shape[0]
        mb = self.micro_batch if 0 < self.micro_batch < N else N
        dense = self._backward_choice(mb, force_naive) if _dense is None else _dense
        self._eager_seen.add(dense)
        loss = None
        hooks = self.model.hooks
        sink_before, hooks.sink = hooks.sink, True
        acc_before = hooks.live_acc
        if hooks.live_acc is None:
            hooks.live_acc = self._live_cum
        try:      # dW kernels add straight into the flat gradient (p.grad are views of it)
            step_prec = self._step_precision(mb, force_naive, dense=dense)
            with self._PrecisionFor(self.model, step_prec):
                self._pack_weights(zero_grad=True)
                for a in range(0, N, mb):
                    b = min(a + mb, N)
                    part = self._fwd_bwd(batch, a, b, i, draws, (b - a) / N, force_naive)
                    loss = part if loss is None else loss + part
            if consist is not None and i >= self.noisenerf_start_iter:
                # computed from i >= noisenerf_start_iter, added to the loss only for i > (run_lushnerf.py:629, 658-659)
                w = 1e-2 if i > self.noisenerf_start_iter else 0.0
                if step_prec is not self.model.precision:
                    self._pack_weights()      # (the aligned-pixel march runs in the model's own kernel variant: other fragment copies)
                loss = loss + w * self._consistency(consist, w)
        finally:
            hooks.sink, hooks.packed, hooks.live_acc = sink_before, None, acc_before
        if not dense and self._live_mode():
            self._live_snapshot(acc_before)      # (the caller's accumulator when it supplied one: the policy sees those steps too)
        if self.distributed:
            self._all_reduce()          # RCCL sum over xGMI; the 1/world mean is folded into Adam
        lr = self.lr() if self._lr_next is None else self._lr_next
        self._lr_next = None
        active = [True, not force_naive, False]
        for s, (a, b) in enumerate(self.flat.segments):
            if active[s] and b > a:
                self.steps[s] += 1
        if self._segments_consecutive and self._adam is ops.adam_step:
            mask = sum(1 << s for s, (a, b) in enumerate(self.flat.segments) if active[s] and b > a)
            ops.adam_step_multi(self.flat.param, self.flat.grad, self.m, self.v, [b for _, b in self.flat.segments], mask, lr, self.steps,
                                grad_scale=1.0 / self.world)
        else:
            for s, (a, b) in enumerate(self.flat.segments):
                if active[s] and b > a:
                    self._adam(self.flat.param[a:b], self.flat.grad[a:b], self.m[a:b], self.v[a:b], lr, self.steps[s],
                               grad_scale=1.0 / self.world)
        self.global_step += 1
        return loss

    # ------------------------------------------------------------------ BASELINE config 1: the coarse-only step
    def step_coarse_only(self, batch: Dict[str, torch.Tensor], draws=None, state: Optional[torch.Tensor] = None):
        """One optimisation step at N_importance = 0 (BASELINE config 1: N_rand 256, 32 + 0, naive).  NeRFAll.forward cannot take
        N_importance = 0 -- the reference indexes extras['rgb0'] there (models/lushnerf.py:660) -- so the entry is render_infer
        (:679-763 -> render_rays :354-479), as in the reference's own CPU-runnable case; the loss is run_lushnerf.py:652-661
        with rgb0 = rgb (ONE summed gradient), Adam on the first segment (coarse network).  What bench.py's C1 figure times and
        what tests/test_gpu_parity.py checks against the oracle, eagerly and as a HIP graph.
        `state` (the device step state, include/lush_march.h lush_step_state_*): the graph-body form -- draw counter, rate and
        Adam step count come from the state and the last node advances it; the host's counters are the caller's to advance
        after a replay.  Returns (loss, tone-mapped colours)."""
        model, hooks = self.model, self.model.hooks
        model.train()
        sink_before, hooks.sink = hooks.sink, True
        if state is not None:
            hooks.state, hooks.draw_delta = state, 0
        try:
            with self._PrecisionFor(model, self._step_precision(batch["target"].shape[0], True, coarse_only=True)):
                self._pack_weights(zero_grad=True)        # one launch: every network re-packed, the flat gradient cleared
                rays = batch["rays"] if "rays" in batch else ops.gen_rays(batch["c2w"], batch["view"], batch["px"], batch["py"], self.K)
                (rgb, depth, acc, extras), noise = model.render_infer(
                    self.H, self.W, self.K, self.chunk, rays=rays, retraw=True, draws=draws, **dict(self.kw, N_importance=0))
                tm = model.tonemapping(rgb)
                loss, g, _ = ops.train_loss_grads(tm, tm, batch["target"], 1.0, work=self._loss_work)      # (a is b: rgb0 = rgb)
                tm.backward(g)
            calls = self._coarse_only_calls = hooks.draw_delta      # lush_draws calls of this step (graph-body form)
        finally:
            hooks.sink, hooks.packed, hooks.state = sink_before, None, None
        a0, a1 = self.flat.segments[0]
        if state is not None:
            ops.adam_step_state(self.flat.param[a0:a1], self.flat.grad[a0:a1], self.m[a0:a1], self.v[a0:a1], state, 0,
                                grad_scale=1.0 / self.world)
            ops.lib.call("lush_step_state_advance", ops.lib.ptr(state), int(calls), 1, float(self.lrate), float(self.lrate_decay * 1000),
                         0.9, 0.999, ops._stream())
        else:
            if self.distributed:
                self._all_reduce()
            lr = self.lr() if self._lr_next is None else self._lr_next
            self._lr_next = None
            self.steps[0] += 1
            ops.adam_step(self.flat.param[a0:a1], self.flat.grad[a0:a1], self.m[a0:a1], self.v[a0:a1], lr, self.steps[0],
                          grad_scale=1.0 / self.world)
            self.global_step += 1
        return loss, tm.detach()

    def capture_coarse_only(self, batch):
        """step_coarse_only captured as ONE HIP graph (call after at least one eager step of the same shapes).  Returns
        (replay, static): static is the graph's own copy of the batch (load the step's inputs into it), replay() launches the
        graph and advances the host's mirror of the counters; replay.loss / replay.tm are the graph's output tensors."""
        if self.world > 1:
            raise NotImplementedError("capture_coarse_only: one rank (the collective is not captured; use step_coarse_only)")
        state = torch.zeros(ops.lib.load().lush_step_state_bytes(), dtype=torch.uint8, device=self.flat.param.device)
        self._state_init(state)
        static = batch.clone_static() if isinstance(batch, BlobBatch) else {k: v.clone() for k, v in batch.items()}
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss, tm = self.step_coarse_only(static, state=state)
        hooks, calls = self.model.hooks, self._coarse_only_calls

        def replay():
            graph.replay()
            self.steps[0] += 1
            self.global_step += 1
            hooks.draw_offset += calls      # (one lush_draws call per step: the march's; the noise branch draws nothing)
        replay.loss, replay.tm, replay.graph, replay.state = loss, tm, graph, state
        return replay, static

    # ------------------------------------------------------------------ the step as one HIP graph
    def _all_reduce(self):
        """The step's ONE collective: sum of the flat gradient over the ranks (RCCL over xGMI; 1/world is folded into Adam)."""
        ev = None
        if self.allreduce_events is not None and self.flat.grad.is_cuda:      # bench.py: HIP events around the one collective
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self.flat.grad)
        if ev is not None:
            ev[1].record()
            self.allreduce_events.append(ev)

    def _body_grads(self, batch, i, force_naive, state, dense=False):
        """First half of what step() enqueues -- zero the flat gradient, forward + backward of every micro-batch -- with the
        Philox draw counter taken from the device step state."""
        N = batch["target"].shape[0]
        mb = self.micro_batch if 0 < self.micro_batch < N else N
        loss = None
        hooks = self.model.hooks
        sink_before, hooks.sink = hooks.sink, True
        hooks.state, hooks.draw_delta = state, 0
        try:
            with self._PrecisionFor(self.model, self._step_precision(mb, force_naive, dense=dense)):
                self._pack_weights(zero_grad=True)
                for a in range(0, N, mb):
                    b = min(a + mb, N)
                    part = self._fwd_bwd(batch, a, b, i, None, (b - a) / N, force_naive)
                    loss = part if loss is None else loss + part
        finally:
            hooks.sink, hooks.state, hooks.packed = sink_before, None, None
        return loss, hooks.draw_delta

    def _body_update(self, force_naive, state, calls):
        """Second half: Adam on the active segments with rate / step counts from the device step state, then advance it."""
        active = [True, not force_naive, False]
        mask = sum(1 << s for s, (a, b) in enumerate(self.flat.segments) if active[s] and b > a)
        ends = [b for _, b in self.flat.segments]
        if self._segments_consecutive:
            ops.adam_step_state_multi(self.flat.param, self.flat.grad, self.m, self.v, ends, state, mask, grad_scale=1.0 / self.world)
        else:
            for s, (a, b) in enumerate(self.flat.segments):
                if active[s] and b > a:
                    ops.adam_step_state(self.flat.param[a:b], self.flat.grad[a:b], self.m[a:b], self.v[a:b], state, s,
                                        grad_scale=1.0 / self.world)
        ops.lib.call("lush_step_state_advance", ops.lib.ptr(state), int(calls), int(mask), float(self.lrate),
                     float(self.lrate_decay * 1000), 0.9, 0.999, ops._stream())
        return mask

    def _state_init(self, state):
        """(Re-)write the device step state from the host's counters: what the graph's kernels read instead of arguments."""
        hooks = self.model.hooks
        steps = (ops.C.c_int * 3)(*self.steps)
        ops.lib.call("lush_step_state_init", ops.lib.ptr(state), ops.C.c_ulonglong(hooks.draw_offset), int(self.global_step), steps,
                     float(self.lrate), float(self.lrate_decay * 1000), 0.9, 0.999, ops._stream())
        return (int(hooks.draw_offset), int(self.global_step), tuple(self.steps))

    def invalidate_graph(self):
        """Forget the captured step (checkpoint load, replica sync: parameters, moments and counters were rewritten)."""
        self._graph = None
        self._graph_eager_left = max(self._graph_eager_left, 1)

    def step_graph(self, batch: Dict[str, torch.Tensor], i: int, split: Optional[bool] = None, consist: Optional[dict] = None):
        """step() with the whole step -- ray generation, both marches, loss, backward, Adam -- captured once in a HIP graph
        and replayed: one graph launch instead of ~50 kernel launches from Python (the launch-bound configurations: BASELINE
        config 1 spends its step in the host's launch path).  What the host passes per step as kernel arguments -- the
        learning rate, Adam's step counts, the Philox draw counter -- lives in the device step state (lush_step_state_*), which
        the graph's last node advances, so replays continue exactly where eager steps would: same draws, same rates.  The host
        keeps a mirror of those counters next to the graph; whenever they no longer match the trainer's own (an eager step()
        in between -- consistency steps, a restored rate --, a checkpoint load, sync_replicas) the state is re-written from
        the host before the replay: no re-capture needed.
        With more than one rank (or split=True) the step is TWO graphs around the one collective: [zero grad, forward,
        backward] -> dist.all_reduce on the same stream, not captured -> [Adam, state advance]; the reference's
        DataParallel semantics (run_lushnerf.py:348, 652-661) as in step().
        The first calls, and one call after every change of the capture key, run step() itself (they are real steps: lazy
        initialisation behind the entry points runs outside a capture); the batch is copied into the graph's static tensors
        before every replay; the returned loss is the graph's static tensor, overwritten by the next replay.  Explicit draws,
        the consistency branch and a rate restored from a checkpoint fall back to step()."""
        force_naive = i < self.kernel_start_iter
        allk = i < self.allkernel_start_iter
        split = (self.world > 1) if split is None else bool(split)
        prec = self.model.precision
        key = (force_naive, allk, split, prec.fwd, prec.bwd, prec.variant,
               tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in batch.items())))
        if self._graph is not None and self._graph["key"] != key:
            self.invalidate_graph()         # (e.g. force_naive flips at kernel_start_iter: the RBK / noise kernels' first call must be eager)
        if self._lr_next is not None or i >= self.noisenerf_start_iter or consist is not None or self._graph_eager_left > 0 \
                or not self.flat.param.is_cuda:
            self._graph_eager_left = max(self._graph_eager_left - 1, 0)
            return self.step(batch, i, consist=consist)      # (the consistency branch draws its anchor / pixels on the host: eager)
        hooks = self.model.hooks
        N = batch["target"].shape[0]
        dense = self._backward_choice(self.micro_batch if 0 < self.micro_batch < N else N, force_naive)
        if dense not in self._eager_seen:             # (each backward's first step is a real, eager one: its lazy initialisation)
            return self.step(batch, i, _dense=dense)
        if self._graph is None or dense not in self._graph["sub"]:
            self.model.train()
            dev = self.flat.param.device
            if self._graph is None:
                state = torch.zeros(ops.lib.load().lush_step_state_bytes(), dtype=torch.uint8, device=dev)
                mirror = self._state_init(state)
                static = batch.clone_static() if isinstance(batch, BlobBatch) else {k: v.clone() for k, v in batch.items()}
                self._graph = dict(key=key, static=static, state=state, mirror=mirror, sub={}, pool=None)
            G = self._graph
            state, static = G["state"], G["static"]
            g1 = torch.cuda.CUDAGraph()
            g2 = None
            pool = {} if G["pool"] is None else dict(pool=G["pool"])      # (the two captured steps never run at once: one pool)
            distributed, self.distributed = self.distributed, False      # (the captured body never calls the collective itself)
            acc_before, hooks.live_acc = hooks.live_acc, (self._live_cum if hooks.live_acc is None else hooks.live_acc)
            try:
                if not split:
                    with torch.cuda.graph(g1, **pool):
                        loss, calls = self._body_grads(static, i, force_naive, state, dense)
                        mask = self._body_update(force_naive, state, calls)
                else:
                    with torch.cuda.graph(g1, **pool):
                        loss, calls = self._body_grads(static, i, force_naive, state, dense)
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, pool=g1.pool()):
                        mask = self._body_update(force_naive, state, calls)
            finally:
                self.distributed, hooks.live_acc = distributed, acc_before
            G["pool"] = g1.pool()
            G["sub"][dense] = dict(g1=g1, g2=g2, loss=loss, calls=calls, mask=mask)
        G = self._graph
        sub = G["sub"][dense]
        if G["mirror"] != (int(hooks.draw_offset), int(self.global_step), tuple(self.steps)):
            G["mirror"] = self._state_init(G["state"])
        if isinstance(batch, BlobBatch) and isinstance(G["static"], BlobBatch):
            G["static"].load(batch)                  # one device copy for all per-step inputs
        else:
            for k, v in batch.items():
                G["static"][k].copy_(v, non_blocking=True)
        sub["g1"].replay()
        if sub["g2"] is not None:
            if self.distributed:
                self._all_reduce()
            sub["g2"].replay()
        if not dense:
            self._live_snapshot()                    # (the captured live-point march adds its counts on the device)
        for s in range(3):
            if (sub["mask"] >> s) & 1:
                self.steps[s] += 1
        self.global_step += 1
        hooks.draw_offset += sub["calls"]
        G["mirror"] = (int(hooks.draw_offset), int(self.global_step), tuple(self.steps))
        return sub["loss"]

    def faults(self) -> int:
        """Numerical-fault word of the model's render calls since the last read (one device sync; the reference
        prints after every chunk, models/lushnerf.py:474-478, 578-582)."""
        return self.model.read_faults()
