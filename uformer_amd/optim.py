"""``AdamW`` on the multi-tensor ``uf_adamw_step`` kernel: same constructor, ``step`` / ``zero_grad`` / ``state_dict`` /
``load_state_dict`` surface and state layout as ``torch.optim.AdamW`` (the reference's optimizer, train/train_denoise.py:77), so
the ``'optimizer'`` entry of a reference checkpoint (train/train_denoise.py:207-235, utils/model_utils.py:50-54) loads and saves
unchanged.  It IS a torch.optim.Optimizer (schedulers such as the reference's warm-up + cosine work on it); only ``step`` is ours:
one launch per 40 parameters instead of ~10 ATen kernels per parameter, f32 state, decoupled weight decay, optional
``grad_scale`` (1 / world_size folds the all-reduce average into the update).

``GradScaler``: the reference trains fp16 under ``torch.cuda.amp.GradScaler`` (through timm's NativeScaler: train/train_denoise.py:42, :180-184 --
dynamic scale, inf / nan check, skipped step on overflow).  Here the whole protocol lives on the device (``uf_grad_scaler_check``,
``uf_adamw_step_scaled``, ``uf_grad_scaler_update``): no host synchronisation per step, the same growth / back-off rule and defaults."""
from __future__ import annotations

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference (train/train_denoise.py:77)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._scaler = None            # the GradScaler whose device-side count currently carries the bias-correction step (None: state['step'] does)

    def _loaded_step(self) -> int:
        return max((int(st["step"]) for st in self.state.values() if "step" in st), default=0)

    def _leave_scaled_mode(self):
        """state['step'] becomes the authority again (one 4-byte device read): before a plain step, and before state_dict()."""
        if self._scaler is not None:
            self._scaler.sync_steps(self)
            self._scaler = None

    def state_dict(self):
        # steps taken under a GradScaler are counted on the device: write them into the per-parameter 'step' entries first, so that the
        # 'optimizer' entry of a checkpoint (train/train_denoise.py:207-235) resumes with the right bias correction under ANY AdamW
        if self._scaler is not None:
            self._scaler.sync_steps(self)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._scaler = None            # the loaded 'step' entries are the authority; the next scaled step seeds the scaler from them

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, scaler: "GradScaler | None" = None):
        if scaler is not None:
            return self._step_scaled(scaler, grad_scale)
        self._leave_scaled_mode()
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

    @torch.no_grad()
    def _step_scaled(self, scaler: "GradScaler", grad_scale: float):
        """The step under a device-resident dynamic loss scale: every parameter shares the scaler's count of steps actually taken (a skipped step does
        not advance it), the per-parameter ``step`` entries of the state dict are refreshed from it by ``GradScaler.sync_steps`` (``state_dict()`` and a
        later plain ``step()`` do that by themselves).  The first scaled step after construction, after ``load_state_dict`` or after plain steps SEEDS the
        scaler's count from the optimizer's own ``step`` entries (when the optimizer has any: a reference checkpoint carries the optimizer but no scaler,
        train/train_denoise.py:207-235): bias corrections continue at step N + 1 instead of restarting at 1 on populated moments."""
        if self._scaler is not scaler:
            loaded = self._loaded_step()
            if loaded > 0:
                # the loaded / plain-step count is the authority (a max() kept a live scaler's LARGER count after the optimizer was rolled back to an
                # earlier checkpoint, and corrupted the saved 'step' entries afterwards); device-side fill, no host synchronisation
                scaler.state[4:5].fill_(float(loaded))
            self._scaler = scaler
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p); gs.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous()); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            if ps:
                ops.adamw_step_scaled(ps, gs, ms, vs, scaler.state, lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"],
                                      grad_scale=grad_scale)
        return None


class GradScaler:
    """``torch.cuda.amp.GradScaler`` for ``uformer_amd.optim.AdamW`` with its state and every decision on the device:

        scaler = GradScaler()                      # init_scale 65536, growth 2, backoff 0.5, growth_interval 2000: torch's defaults
        loss = criterion(model(x), target)
        scaler.scale(loss).backward()
        scaler.step(optimizer)                     # inf / nan check over the gradients, unscale folded into the update, skipped on overflow
        scaler.update()

    (train/train_denoise.py:180-184 does the same through timm's NativeScaler.)  ``get_scale()`` / ``steps_taken()`` read the device state (one
    4-byte copy: for logging, not for the training loop)."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000, device="cuda"):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.tensor([init_scale, 1.0 / init_scale, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self.state[0]

    @torch.no_grad()
    def step(self, optimizer: AdamW, grad_scale: float = 1.0):
        grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        ops.grad_scaler_check([g if g.is_contiguous() else g.contiguous() for g in grads], self.state)
        return optimizer.step(grad_scale=grad_scale, scaler=self)

    def update(self):
        ops.grad_scaler_update(self.state, self.growth_factor, self.backoff_factor, self.growth_interval)

    def get_scale(self) -> float:
        return float(self.state[0].item())

    def steps_taken(self) -> int:
        return int(self.state[4].item())

    def sync_steps(self, optimizer: AdamW):
        """Write the count of steps actually taken into the optimizer's per-parameter ``step`` entries (what ``state_dict()`` / a reference checkpoint
        carries, utils/model_utils.py:50-54) -- call before saving."""
        k = float(self.steps_taken())
        for st in optimizer.state.values():
            if "step" in st:
                st["step"] = torch.tensor(k)

    def state_dict(self):
        s = self.state.cpu()
        return {"scale": float(s[0]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(s[3]), "_steps_taken": int(s[4])}

    def load_state_dict(self, d):
        self.growth_factor, self.backoff_factor, self.growth_interval = d["growth_factor"], d["backoff_factor"], d["growth_interval"]
        self.state.copy_(torch.tensor([d["scale"], 1.0 / d["scale"], 0.0, float(d.get("_growth_tracker", 0)), float(d.get("_steps_taken", 0)), 0.0, 0.0, 0.0]))
