"""``AdamW`` on the multi-tensor ``uf_adamw_step`` kernel: same constructor, ``step`` / ``zero_grad`` / ``state_dict`` /
``load_state_dict`` surface and state layout as ``torch.optim.AdamW`` (the reference's optimizer, train/train_denoise.py:77), so
the ``'optimizer'`` entry of a reference checkpoint (train/train_denoise.py:207-235, utils/model_utils.py:50-54) loads and saves
unchanged.  It IS a torch.optim.Optimizer (schedulers such as the reference's warm-up + cosine work on it); only ``step`` is ours:
one launch per 40 parameters instead of ~10 ATen kernels per parameter, f32 state, decoupled weight decay, optional
``grad_scale`` (1 / world_size folds the all-reduce average into the update)."""
from __future__ import annotations

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference (train/train_denoise.py:77)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                k = int(st["step"])
                if step is None:
                    step = k
                if k != step or p.dtype != torch.float32:   # mixed step counts (a parameter that skipped steps): one launch per count
                    ops.adamw_step([p], [p.grad.contiguous()], [st["exp_avg"]], [st["exp_avg_sq"]], lr=group["lr"], betas=group["betas"],
                                   eps=group["eps"], weight_decay=group["weight_decay"], step=k, grad_scale=grad_scale)
                    continue
                ps.append(p); gs.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous()); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            if ps:
                ops.adamw_step(ps, gs, ms, vs, lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"],
                               step=step, grad_scale=grad_scale)
        return loss
