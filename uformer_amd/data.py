"""Training input pipeline on the device.  The reference decodes PNGs and crops / rotates on the CPU per sample
(dataset/dataset_denoise.py:42-73) and mixes on the GPU (utils/dataset_utils.py:37-53).  Here the decoded frames live in HBM as
uint8 (288 GB holds every SIDD-Medium / GoPro training frame) and one kernel per batch produces the f32 patches; the random
choices are drawn on the host with the reference's own distributions: ``np.random.randint(0, H - ps)`` for the corner,
``random.getrandbits(3)`` for the transform, ``torch.randperm`` + Beta(1.2, 1.2) for MixUp."""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops

# transforms_aug = sorted(dir(Augment_RGB_torch)) non-underscore callables = transform0..transform7 in this order
N_TRANSFORMS = 8


def sample_patch_meta(batch: int, n_frames: int, H: int, W: int, ps: int, indices=None, np_rng=np.random, py_rng=random) -> torch.Tensor:
    """(B,4) int32 [frame, r0, c0, transform]: corner = np.random.randint(0, H - ps) (0 when H == ps: dataset_denoise.py:58-63),
    transform = random.getrandbits(3) (:67)."""
    rows = []
    for b in range(batch):
        idx = int(indices[b]) if indices is not None else py_rng.randrange(n_frames)
        if H - ps == 0:
            r = c = 0
        else:
            r = int(np_rng.randint(0, H - ps))
            c = int(np_rng.randint(0, W - ps))
        rows.append((idx % n_frames, r, c, py_rng.getrandbits(3)))
    return torch.tensor(rows, dtype=torch.int32)


class GpuPatchLoader:
    """clean / noisy frame stacks resident on the GPU ((N,H,W,3) uint8 as decoded, or (N,3,H,W) f32) -> training batches."""

    def __init__(self, clean: torch.Tensor, noisy: torch.Tensor, patch_size: int, hwc: Optional[bool] = None):
        if clean.shape != noisy.shape:
            raise ValueError("clean / noisy stacks differ in shape")
        self.hwc = (clean.shape[-1] == 3 and clean.shape[1] != 3) if hwc is None else hwc
        self.clean, self.noisy, self.ps = clean, noisy, patch_size
        self.N = clean.shape[0]
        self.H, self.W = (clean.shape[1], clean.shape[2]) if self.hwc else (clean.shape[2], clean.shape[3])

    def batch(self, batch_size: int, indices=None) -> Tuple[torch.Tensor, torch.Tensor]:
        meta = sample_patch_meta(batch_size, self.N, self.H, self.W, self.ps, indices).to(self.clean.device, non_blocking=True)
        return ops.crop_augment(self.clean, meta, self.ps, self.hwc), ops.crop_augment(self.noisy, meta, self.ps, self.hwc)


class MixUp_AUG:
    """utils/dataset_utils.py:37-53."""

    def __init__(self):
        self.dist = torch.distributions.beta.Beta(torch.tensor([1.2]), torch.tensor([1.2]))

    def aug(self, rgb_gt: torch.Tensor, rgb_noisy: torch.Tensor):
        bs = rgb_gt.size(0)
        indices = torch.randperm(bs)
        lam = self.dist.rsample((bs, 1)).view(-1)
        dev = rgb_gt.device
        perm, lam = indices.to(dev, torch.int32), lam.to(dev)
        return ops.mixup(rgb_gt, lam, perm), ops.mixup(rgb_noisy, lam, perm)
