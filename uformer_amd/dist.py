"""Multi-GPU plumbing for the hot path: one process per GPU, batch sharding, no data-path
collective for inference (images never interact, weights are replicated).

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(train/train_denoise.py:83); here each rank owns ``shard_batch`` of the global batch and the
ranks meet only to agree on the wall-clock (max over ranks) and, optionally, to gather outputs.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str) -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of the global batch owned by ``rank``; sizes differ by <= 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device="cpu") -> float:
    """Slowest rank's value (the job's wall-clock)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_batch(local: torch.Tensor, global_batch: int) -> torch.Tensor:
    """All ranks' output shards concatenated in rank order (evaluation convenience; not on the
    timed path).  Shards may differ by one image, so they are padded to the largest shard."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_batch(global_batch, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: b - a] for r, (a, b) in enumerate(sizes)], 0)


# ---------------------------------------------------------------------------------------------------------------------
# training: the one exchange step of data parallelism -- average the gradients over the ranks.
# Replaces nn.DataParallel's gather-on-GPU-0 (train/train_denoise.py:83) by a bucketed all-reduce over RCCL (backend
# "nccl" on ROCm) / gloo.
# ---------------------------------------------------------------------------------------------------------------------
class GradientAllReduce:
    """Bucketed mean of ``param.grad`` over all ranks.

    Parameters are laid out in REVERSE registration order (the order the backward sweep finishes them: decoder first) into
    flat f32 buckets of about ``bucket_bytes`` (25 MB: large enough to run at xGMI link rate, small enough that several are
    in flight); every bucket is one asynchronous all-reduce.  Gradients that are None on this rank (a block whose two
    branches DropPath dropped for every local sample) take part as zeros -- the collective must be the same on all ranks.
    203.5 MB of fp32 gradients for Uformer-B = 9 buckets."""

    def __init__(self, params, bucket_bytes: int = 25 << 20):
        self.params = [p for p in params if p.requires_grad][::-1]
        self.buckets = []            # lists of parameter indices
        cur, size = [], 0
        for i, p in enumerate(self.params):
            n = p.numel() * 4
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += n
        if cur:
            self.buckets.append(cur)
        self._flat = None

    def _buffers(self, device):
        if self._flat is None or self._flat[0].device != device:
            self._flat = [torch.empty(sum(self.params[i].numel() for i in b), dtype=torch.float32, device=device) for b in self.buckets]
        return self._flat

    def __call__(self) -> None:
        """All-reduce and average in place; a no-op in a single-process run."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        flats = self._buffers(self.params[0].device)
        works = []
        for b, flat in zip(self.buckets, flats):
            off = 0
            for i in b:
                p = self.params[i]
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))     # later buckets pack while this one flies
        for b, flat, w in zip(self.buckets, flats, works):
            w.wait()
            flat.div_(world)
            off = 0
            for i in b:
                p = self.params[i]
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
