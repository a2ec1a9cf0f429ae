"""Training step around the HIP ray march: the build's counterpart of the reference's
train() inner loop (run_lushnerf.py:603-685) and of its nn.DataParallel wrapper (:348).

* loss = 0.5*MSE + 0.5*L1 on rgb_blur and rgb0_blur (:652-661)
* Adam(lr 5e-4), lr = lrate * 0.1**(global_step / (lrate_decay*1000)) (:368-371, :681-685)
* data parallel: one process per GPU, each draws its own N_rand rays; ONE all-reduce
  (RCCL over xGMI) of a single flat fp32 gradient buffer per step, then Adam runs
  redundantly on every rank.  No other collective.

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


class Trainer:
    def __init__(self, model: NeRFAll, H: int, W: int, focal: float, N_samples: int = 64, N_importance: int = 64,
                 lrate: float = 5e-4, lrate_decay: int = 250, perturb: float = 1., raw_noise_std: float = 1.,
                 kernel_start_iter: int = 0, allkernel_start_iter: int = 0, chunk: int = 1024 * 32,
                 distributed: bool = False, micro_batch: int = 0):
        self.model = model
        self.H, self.W = H, W
        self.K = [[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]]
        self.kw = dict(perturb=perturb, N_importance=N_importance, N_samples=N_samples, use_viewdirs=True,
                       white_bkgd=False, raw_noise_std=raw_noise_std, inference=False, near=0., far=1.)
        self.lrate, self.lrate_decay = lrate, lrate_decay
        self.kernel_start_iter, self.allkernel_start_iter = kernel_start_iter, allkernel_start_iter
        self.chunk = chunk
        # micro_batch > 0: forward+backward run per slice of that many INPUT rays and gradients accumulate in
        # the flat buffer (the loss is a mean over rays, so this is exact up to summation order).  It bounds the
        # activation stash: BASELINE config 5 (16 384 rays, 128+128) would otherwise hold ~260 GB between
        # forward and backward.
        self.micro_batch = micro_batch
        self.distributed = distributed and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.distributed else 1
        base = list(model.mlp_coarse.parameters()) + (list(model.mlp_fine.parameters()) if model.mlp_fine else [])
        dead = list(model.mlp_noise_coarse.alpha_linear.parameters())
        dead_ids = {id(p) for p in dead}
        late = [p for p in model.mlp_noise_coarse.parameters() if id(p) not in dead_ids]
        if model.blur_kernel_net is not None:
            late = list(model.blur_kernel_net.parameters()) + late
        self.flat = FlatParams([base, late, dead])
        n = self.flat.numel
        self.m = torch.zeros(n, dtype=torch.float32, device=self.flat.param.device)
        self.v = torch.zeros_like(self.m)
        self.steps = [0, 0, 0]            # per-segment Adam step counters (torch keeps one per parameter)
        self.global_step = 0

    def lr(self) -> float:
        return self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 1000)))

    def step(self, batch: Dict[str, torch.Tensor], i: int, draws=None):
        """One optimisation step on a batch {rays [N,3,2], images_idx [N,1], target [N,3], fq_mask [N]}."""
        self.model.train()
        self.flat.grad.zero_()
        force_naive = i < self.kernel_start_iter
        N = batch["rays"].shape[0]
        mb = self.micro_batch if 0 < self.micro_batch < N else N
        M = 1 if force_naive else self.model.mlp_rbk.num_motion + 1
        loss = None
        for a in range(0, N, mb):
            b = min(a + mb, N)
            d = None if draws is None else {k: v[a * M:b * M] for k, v in draws.items()}
            out = self.model(self.H, self.W, self.K, chunk=self.chunk, rays=batch["rays"][a:b],
                             rays_info={"images_idx": batch["images_idx"][a:b]}, retraw=True,
                             force_naive=force_naive, allkernel=i < self.allkernel_start_iter,
                             kernel_pixel=batch["fq_mask"][a:b], draws=d, **self.kw)
            part = ops.TrainLoss.apply(out[0], out[1], batch["target"][a:b]) * ((b - a) / N)
            part.backward()
            loss = part.detach() if loss is None else loss + part.detach()
        if self.distributed:
            dist.all_reduce(self.flat.grad)          # RCCL sum over xGMI; the 1/world mean is folded into Adam
        lr = self.lr()
        active = [True, not force_naive, False]
        for s, (a, b) in enumerate(self.flat.segments):
            if active[s] and b > a:
                self.steps[s] += 1
                ops.adam_step(self.flat.param[a:b], self.flat.grad[a:b], self.m[a:b], self.v[a:b], lr, self.steps[s],
                              grad_scale=1.0 / self.world)
        self.global_step += 1
        return loss
