"""Checkpoint interop with the reference's .tar files.

The reference saves ``{'global_step', 'network_state_dict', 'optimizer_state_dict'}`` where the
network is an ``nn.DataParallel(NeRFAll)`` -- every key carries a ``module.`` prefix and the RBK
appears under three aliases (run_lushnerf.py:687-694; models/lushnerf.py:184, 219-220) -- and
reloads with ``smart_load_state_dict`` (utils/run_lushnerf_helpers.py:612-628: strip 7 characters,
``strict=False``).  The optimizer is ``torch.optim.Adam`` over two parameter groups: everything
except the noise MLP, then the noise MLP (run_lushnerf.py:359-371).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .model import NeRFAll


def reference_state_dict(model: NeRFAll) -> Dict[str, torch.Tensor]:
    """What ``nn.DataParallel(nerf).state_dict()`` holds in the reference."""
    return {"module." + k: v.detach().clone() for k, v in model.state_dict().items()}


def _param_groups(model: NeRFAll):
    noise_ids = {id(p) for p in model.mlp_noise_coarse.parameters()}
    base = [p for p in model.parameters() if id(p) not in noise_ids]
    return base, list(model.mlp_noise_coarse.parameters())


def _group(lr: float, ids) -> dict:
    return {"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
            "foreach": None, "capturable": False, "differentiable": False, "fused": None,
            "decoupled_weight_decay": False, "params": list(ids)}


def adam_state_to_reference(model: NeRFAll, trainer=None, lrate: float = 5e-4) -> dict:
    """``torch.optim.Adam.state_dict()`` in the reference's layout: two parameter groups -- everything except the
    noise MLP, then the noise MLP (run_lushnerf.py:359-371) -- with consecutive parameter indices.  With a trainer
    its flat Adam moments fill ``state``; without one the state is empty (a freshly constructed optimizer), which
    the reference's unconditional ``optimizer.load_state_dict`` (:386) accepts."""
    base, noise = _param_groups(model)
    state, idx, groups = {}, 0, []
    # the rate the reference's optimizer holds when it saves: set from the un-incremented global_step (:681-694)
    lr = lrate if trainer is None else trainer.lrate * (0.1 ** (trainer.global_step / (trainer.lrate_decay * 1000)))
    for params in (base, noise):
        ids = []
        for p in params:
            if trainer is not None:
                flat = trainer.flat
                off = (p.data_ptr() - flat.param.data_ptr()) // 4
                seg = next(s for s, (a, b) in enumerate(flat.segments) if a <= off < b)
                step = trainer.steps[seg]
                if step > 0:            # torch keeps no state for parameters that never received a gradient
                    n = p.numel()
                    state[idx] = {"step": torch.tensor(float(step)),
                                  "exp_avg": trainer.m[off:off + n].view_as(p).detach().clone(),
                                  "exp_avg_sq": trainer.v[off:off + n].view_as(p).detach().clone()}
            ids.append(idx)
            idx += 1
        groups.append(_group(lr, ids))
    return {"state": state, "param_groups": groups}


def load_adam_state(trainer, opt_state: dict):
    """Inverse of adam_state_to_reference (accepts a reference checkpoint's optimizer_state_dict).  Moments of
    parameters the file has no state for are cleared, as a fresh torch optimizer would hold none."""
    flat = trainer.flat
    base, noise = _param_groups(trainer.model)
    params = base + noise
    seg_steps = [0] * len(flat.segments)
    trainer.m.zero_()
    trainer.v.zero_()
    for idx, st in opt_state["state"].items():
        p = params[int(idx)]
        off = (p.data_ptr() - flat.param.data_ptr()) // 4
        n = p.numel()
        trainer.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(trainer.m.device))
        trainer.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(trainer.v.device))
        seg = next(s for s, (a, b) in enumerate(flat.segments) if a <= off < b)
        seg_steps[seg] = max(seg_steps[seg], int(float(st["step"])))
    trainer.steps = seg_steps
    groups = opt_state.get("param_groups") or []
    if groups and "lr" in groups[0]:
        # the reference resumes with the rate stored in the optimizer (torch restores param_group['lr']) and only
        # recomputes it after its first step (run_lushnerf.py:386, 681-685)
        trainer._lr_next = float(groups[0]["lr"])


def save_checkpoint(path: str, model: NeRFAll, global_step: int, trainer=None):
    """Write a file the reference's loader (run_lushnerf.py:373-389) accepts."""
    ck = {"global_step": int(global_step), "network_state_dict": reference_state_dict(model),
          "optimizer_state_dict": adam_state_to_reference(model, trainer)}
    if trainer is not None and hasattr(trainer, "live_policy_state"):
        # one key the reference's loader never looks at (it reads the three above): where the live-point policy stands, so that a
        # resumed run makes the choices the uninterrupted one would have made (trainer.py, "Policy")
        ck["lush_live_policy"] = trainer.live_policy_state()
    torch.save(ck, path)


def load_checkpoint(path: str, model: NeRFAll, trainer=None, map_location="cpu") -> int:
    """smart_load_state_dict semantics (helpers:612-628): strip the 7-character 'module.' prefix,
    load non-strictly; returns global_step."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = {k[7:]: v for k, v in ck["network_state_dict"].items()}
    own = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                own[k].copy_(v.to(own[k].device))       # in place: parameters may be views of the flat buffer
    if trainer is not None:
        trainer.global_step = int(ck.get("global_step", 0))       # start = ckpt['global_step'] (run_lushnerf.py:385, 419)
        load_adam_state(trainer, ck.get("optimizer_state_dict") or {"state": {}, "param_groups": []})
        if hasattr(trainer, "load_live_policy_state"):
            trainer.load_live_policy_state(ck.get("lush_live_policy"))   # (absent in a reference-written file: the policy starts over)
        trainer.sync_replicas()                                   # a file read on one rank must not fork the replicas
        trainer.invalidate_graph()                                # a captured step's device counters / rate are stale now
    return int(ck.get("global_step", 0))
