"""Training criterion of the reference (losses.py:41-52 ``CharbonnierLoss``, used at train/train_denoise.py:164,181) on the fused
``uf_charbonnier_fwd_bwd`` kernel: the forward pass computes the loss AND d loss / d restored in one sweep over the image, so
``loss.backward()`` costs nothing extra before it enters the model's reverse sweep."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class _Charbonnier(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        loss, dx = ops.charbonnier(x.detach(), y.detach(), eps, with_grad=x.requires_grad or y.requires_grad)
        ctx.save_for_backward(dx)
        ctx.need = (x.requires_grad, y.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        gx = dx * g if ctx.need[0] else None
        gy = -dx * g if ctx.need[1] else None
        return gx, gy, None


class CharbonnierLoss(nn.Module):
    """Charbonnier Loss (L1): mean(sqrt((x - y)^2 + eps^2)).  Same constructor / call as the reference's."""

    def __init__(self, eps: float = 1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return _Charbonnier.apply(x, y, self.eps)
