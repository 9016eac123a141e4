"""Evaluation metrics of the reference on the device: no image leaves HBM, one float per image comes back.
``myPSNR`` / ``batch_PSNR``: utils/image_utils.py:40-51; ``batch_SSIM``: utils/caculate_psnr_ssim.py:35-81 (uint8-quantised, 11x11
Gaussian window, valid region)."""
from __future__ import annotations

import torch

from . import ops


def myPSNR(tar_img: torch.Tensor, prd_img: torch.Tensor) -> torch.Tensor:
    """20 log10(1 / rmse) of the [0,1]-clamped images; (C,H,W) or (1,C,H,W)."""
    a = tar_img if tar_img.dim() == 4 else tar_img.unsqueeze(0)
    b = prd_img if prd_img.dim() == 4 else prd_img.unsqueeze(0)
    return 20 * torch.log10(1 / ops.batch_mse(b, a).sqrt()).reshape(())


def batch_PSNR(img1: torch.Tensor, img2: torch.Tensor, average: bool = True) -> torch.Tensor:
    """Per-image myPSNR summed (or averaged) over the batch, as the validation loop uses it (train/train_denoise.py:197)."""
    ps = 20 * torch.log10(1 / ops.batch_mse(img1, img2).sqrt())
    return ps.mean() if average else ps.sum()


def batch_SSIM(img1: torch.Tensor, img2: torch.Tensor, average: bool = True) -> torch.Tensor:
    s = ops.batch_ssim(img1, img2)
    return s.mean() if average else s
