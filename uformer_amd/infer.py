"""Arbitrary-resolution inference wrapper: the pad -> forward -> crop -> clamp steps the reference's evaluation scripts
put around ``Uformer.forward`` (test/test_sidd.py:79-92 ``expand2square``, :106-109; test/test_gopro_hide.py:77-103),
kept on the device the image lives on.  Glue only (a handful of torch slice copies per image); the hot path is the model.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

Tensor = torch.Tensor


def expand2square(timg: Tensor, factor: float = 128.0) -> Tuple[Tensor, Tensor]:
    """(B,C,h,w) -> zero canvas (B,C,X,X) with the image centred at ((X-h)//2, (X-w)//2), X = max(h,w) rounded up to a
    multiple of ``factor`` (128 = 4 down-samplings x window 8), and the (B,1,X,X) mask of ones over the image.
    Same arithmetic as the reference helper (which is written for B = 1), on ``timg``'s device."""
    b, c, h, w = timg.shape
    X = int(math.ceil(max(h, w) / float(factor)) * factor)
    y0, x0 = (X - h) // 2, (X - w) // 2
    img = torch.zeros(b, c, X, X, dtype=timg.dtype, device=timg.device)
    mask = torch.zeros(b, 1, X, X, dtype=timg.dtype, device=timg.device)
    img[:, :, y0:y0 + h, x0:x0 + w] = timg
    mask[:, :, y0:y0 + h, x0:x0 + w] = 1
    return img, mask


def crop_to_mask(restored: Tensor, h: int, w: int) -> Tensor:
    """Inverse of expand2square for the restored canvas: the (B,C,h,w) region the mask covers
    (``torch.masked_select(restored, mask.bool()).reshape(1,3,h,w)`` in the reference, as a slice)."""
    X = restored.shape[-1]
    y0, x0 = (X - h) // 2, (X - w) // 2
    return restored[:, :, y0:y0 + h, x0:x0 + w]


@torch.no_grad()
def restore(model, img: Tensor, factor: float = 128.0, clamp: bool = True) -> Tensor:
    """Restore images of any (h, w): pad to a square multiple of ``factor``, run ``model``, crop back, clamp to [0,1]
    (test/test_sidd.py:106-109).  ``img``: (B,3,h,w) float32 on the model's device."""
    if img.dim() != 4:
        raise ValueError(f"restore expects (B,C,h,w), got {tuple(img.shape)}")
    h, w = img.shape[-2:]
    padded, _ = expand2square(img, factor)
    out = crop_to_mask(model(padded), h, w)
    return torch.clamp(out, 0, 1) if clamp else out
