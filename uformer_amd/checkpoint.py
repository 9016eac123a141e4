"""Checkpoints in the reference's dict format (train/train_denoise.py:207-235: ``{'epoch', 'state_dict', 'optimizer'}``;
utils/model_utils.py:18-54 ``save_checkpoint`` / ``load_checkpoint`` / ``load_start_epoch`` / ``load_optim``), so a run can be
resumed by either code base."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch


def save_checkpoint(model_dir: str, state: dict, session: str) -> str:
    """utils/model_utils.py:18-21."""
    path = os.path.join(model_dir, "model_epoch_{}_{}.pth".format(state["epoch"], session))
    torch.save(state, path)
    return path


def save_training_state(path: str, epoch: int, model, optimizer, data_parallel_prefix: bool = False) -> None:
    """The dict the reference's training loop writes (train/train_denoise.py:207-235).  ``data_parallel_prefix`` adds the
    ``module.`` prefix an nn.DataParallel-wrapped reference model would have produced."""
    sd = model.state_dict()
    if data_parallel_prefix:
        sd = OrderedDict(("module." + k, v) for k, v in sd.items())
    torch.save({"epoch": epoch, "state_dict": sd, "optimizer": optimizer.state_dict()}, path)


def load_checkpoint(model, weights: str, map_location="cpu") -> None:
    """utils/model_utils.py:23-33 (strips ``module.``)."""
    checkpoint = torch.load(weights, map_location=map_location)
    state_dict = checkpoint["state_dict"]
    model.load_state_dict(OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items()))


def load_start_epoch(weights: str) -> int:
    """utils/model_utils.py:45-48."""
    return torch.load(weights, map_location="cpu")["epoch"]


def load_optim(optimizer, weights: str) -> float:
    """utils/model_utils.py:50-54: restores the optimizer state, returns its learning rate."""
    checkpoint = torch.load(weights, map_location="cpu")
    optimizer.load_state_dict(checkpoint["optimizer"])
    lr = None
    for p in optimizer.param_groups:
        lr = p["lr"]
    return lr
