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


def rank_devices(device="cpu"):
    """Proof that N ranks met: one row per rank -- (rank, local rank, device index, 31-bit hash of the device's identity) -- gathered
    over the process group's own backend (RCCL for "nccl").  Rank order; a single process returns its own row."""
    row = [0, 0, -1, 0]
    rank, local_rank, _ = env_rank_world()
    row[0], row[1] = rank, local_rank
    dev = torch.device(device)
    if dev.type == "cuda":
        import hashlib
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(idx)
        ident = f"{pr.name}|{getattr(pr, 'uuid', '')}|{getattr(pr, 'pci_bus_id', '')}|{getattr(pr, 'pci_device_id', '')}|{idx}"
        row[2], row[3] = idx, int.from_bytes(hashlib.sha256(ident.encode()).digest()[:4], "little") & 0x7fffffff
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [row]
    t = torch.tensor(row, dtype=torch.int64, device=device)
    rows = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, t)
    return [[int(v) for v in r.tolist()] for r in rows]


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


class OverlappedGradientAllReduce:
    """The gradient exchange overlapped with the reverse sweep.

    * ``param.grad`` of every parameter IS a view into a flat f32 bucket (about ``bucket_bytes`` each, parameters in reverse
      registration order = the order the reverse sweep finishes them): nothing is packed or unpacked around the collective.
    * the training tape (uformer_amd/train.py ``UformerTape.backward``) calls ``deliver`` with the gradients of every stage as soon
      as the sweep has finished it; they are written into their bucket views, and a bucket whose last gradient has arrived is
      all-reduced at once with ``async_op=True`` -- RCCL runs it on its own stream behind the producing kernels, while the compute
      stream goes on with the next stage's backward.
    * the sum is NOT divided: ``grad_scale`` (1 / world) is handed to ``uformer_amd.optim.AdamW.step`` which folds it into the
      update (``uf_adamw_step``), so the averaged gradient never makes an extra pass through HBM.
    Use: ``sink = OverlappedGradientAllReduce(model); model.grad_sink = sink`` then per step
    ``sink.begin_step(); loss.backward(); sink.finish(); opt.step(grad_scale=sink.grad_scale)``.  ``begin_step`` may be omitted
    (the first gradient after ``finish`` starts a new step); ``begin_step(accumulate=k)`` announces k backward passes per step."""

    def __init__(self, model_or_named_params, bucket_bytes: int = 25 << 20, algorithm: str = "ring", payload: torch.dtype = torch.float32):
        """``algorithm``: "ring" = one ``dist.all_reduce`` per bucket (RCCL / gloo pick the schedule: a ring over xGMI is bound by ONE link per hop);
        "direct" (xGMI is a full point-to-point mesh, 7 links per GPU) = reduce-scatter by ``all_to_all_single`` -- every rank sends
        slice j of the bucket straight to rank j over its own link -- an f32 sum of the world slices on the receiver, and ``all_gather_into_tensor``
        of the reduced slices: two steps over all links at once instead of 2 (world - 1) ring hops.
        ``payload`` (direct only): the wire type.  torch.bfloat16 halves the bytes on the links (101.8 MB instead of 203.5 MB per step for Uformer-B);
        the slices are ACCUMULATED IN F32 on receipt and the reduced slice is rounded once more for the gather, identically on every rank (the
        owner of a slice uses the rounded values too: replicas stay bit-identical)."""
        if algorithm not in ("ring", "direct"):
            raise ValueError("algorithm must be 'ring' or 'direct'")
        if payload not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("payload must be float32, bfloat16 or float16")
        if payload != torch.float32 and algorithm != "direct":
            raise ValueError("a 2-byte payload needs algorithm='direct' (a ring all-reduce would ACCUMULATE in the wire type)")
        self.algorithm, self.payload = algorithm, payload
        named = list(model_or_named_params.named_parameters()) if hasattr(model_or_named_params, "named_parameters") else list(model_or_named_params)
        named = [(n, p) for n, p in named if p.requires_grad][::-1]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.bucket_of, self.buckets = {}, []
        cur, size = [], 0
        for i, p in enumerate(self.params):
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat, self.views = [], {}
        for b, idxs in enumerate(self.buckets):
            dev = self.params[idxs[0]].device
            n_el = sum(self.params[i].numel() for i in idxs)
            n_pad = (n_el + 255) // 256 * 256           # "direct": the bucket splits into `world` equal slices (any world that divides 256: 1, 2, 4, 8, ...; others pad at launch)
            flat = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            self.flat.append(flat)
            off = 0
            for i in idxs:
                p = self.params[i]
                v = flat[off:off + p.numel()].view_as(p)
                self.views[self.names[i]] = v
                self.bucket_of[self.names[i]] = b
                p.grad = v                                         # the optimizer reads the bucket directly
                off += p.numel()
        self.launch_order = []                                     # bucket indices in the order their collectives were issued (tests)
        self._stage = {}                                           # "direct": per-bucket staging buffers (send / receive / reduced slice / gathered), made once
        self._xs = None                                            # "direct" on a GPU: the stream the two-step exchange is chained on
        self.begin_step()

    @property
    def world(self) -> int:
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def begin_step(self, accumulate: int = 1) -> None:
        """Start a step of ``accumulate`` backward passes (gradient accumulation): the passes add into the bucket views and the
        collectives are launched during the LAST one.  Optional for ``accumulate == 1``: the first ``deliver`` after ``finish``
        begins a new step by itself, so a loop written for the reference (zero_grad / backward / step) cannot silently reuse
        the previous step's gradients."""
        if accumulate < 1:
            raise ValueError("accumulate must be >= 1")
        self._accumulate = accumulate
        self._pass = 0
        self._missing = [len(b) for b in self.buckets]
        self._seen = set()                                         # names delivered in the current pass
        self._ever = set()                                         # names delivered in any pass of this step
        self._launched = set()
        self._works = []
        self.launch_order = []
        self._finished = False

    def _launch(self, b: int) -> None:
        self._launched.add(b)
        self.launch_order.append(b)
        if self.world > 1:
            if self.algorithm == "direct":
                self._direct(b)
            else:
                self._works.append(dist.all_reduce(self.flat[b], op=dist.ReduceOp.SUM, async_op=True))

    def wire_bytes_per_step(self) -> int:
        """bytes one rank puts on its links per step (both directions count once): ring all-reduce 2 (w-1)/w N 4, direct 2 (w-1)/w N sizeof(payload)"""
        w = self.world
        if w <= 1:
            return 0
        n = sum(int(f.numel()) for f in self.flat)
        return int(2 * (w - 1) / w * n * (4 if self.algorithm == "ring" else torch.empty(0, dtype=self.payload).element_size()))

    def _direct(self, b: int) -> None:
        """reduce-scatter by all-to-all + f32 accumulation on receipt + all-gather of the reduced slices, chained on a side stream (GPU) so that
        the compute stream goes on with the reverse sweep; finish() joins the stream"""
        flat, w = self.flat[b], self.world
        n = flat.numel()
        npad = (n + w - 1) // w * w
        st = self._stage.get(b)
        if st is None or st["w"] != w:
            st = self._stage[b] = dict(w=w, send=torch.zeros(npad, dtype=self.payload, device=flat.device), recv=torch.empty(npad, dtype=self.payload, device=flat.device),
                                       mine=torch.empty(npad // w, dtype=self.payload, device=flat.device), out=torch.empty(npad, dtype=self.payload, device=flat.device))

        def exchange():
            st["send"][:n].copy_(flat)                                             # f32 -> payload (a plain copy for f32)
            dist.all_to_all_single(st["recv"], st["send"])                         # recv[j] = rank j's copy of MY slice
            red = st["recv"].view(w, npad // w).to(torch.float32).sum(0)          # accumulated in f32 whatever the wire type
            st["mine"].copy_(red)                                                  # rounded ONCE to the wire type: every rank, the owner included, uses these values
            dist.all_gather_into_tensor(st["out"], st["mine"])
            flat.copy_(st["out"][:n])                                              # back into the bucket param.grad views

        if flat.is_cuda:
            if self._xs is None:
                self._xs = torch.cuda.Stream(device=flat.device)
            self._xs.wait_stream(torch.cuda.current_stream(flat.device))           # behind the kernels that produced the bucket's last gradient
            with torch.cuda.stream(self._xs):
                exchange()                                                         # RCCL enqueues on its own stream; this stream waits for it (no host block)
        else:
            exchange()

    def deliver(self, grads) -> None:
        """grads: {parameter name: gradient tensor or None}.  None (a branch DropPath removed on this rank) counts as zeros: the
        collective must be the same on all ranks.  A name that arrives again starts the next backward pass of the step: its
        gradient is ADDED; more passes than ``begin_step(accumulate=...)`` announced is an error (the buckets of the last
        announced pass are already being reduced)."""
        if self._finished:
            self.begin_step()                                      # a new step began without begin_step(): never reuse stale state
        for name, g in grads.items():
            if name not in self.views:
                continue
            if name in self._seen:                                 # second sighting: a new backward pass
                self._pass += 1
                if self._pass >= self._accumulate:
                    raise RuntimeError(f"OverlappedGradientAllReduce: gradient of {name!r} delivered in backward pass {self._pass + 1} of a step announced "
                                       f"with accumulate={self._accumulate}; call finish() and the optimizer step between backwards, or "
                                       f"begin_step(accumulate=k) for k backward passes per step")
                self._seen = set()
                self._missing = [len(b) for b in self.buckets]
            self._seen.add(name)
            v = self.views[name]
            first = name not in self._ever
            self._ever.add(name)
            if g is None:
                if first:
                    v.zero_()
            elif first:
                if g.data_ptr() != v.data_ptr():
                    v.copy_(g.reshape(v.shape))
            else:
                v.add_(g.reshape(v.shape))
            b = self.bucket_of[name]
            self._missing[b] -= 1
            if self._missing[b] == 0 and self._pass == self._accumulate - 1:
                self._launch(b)

    def delivered(self, names) -> bool:
        """True when every one of ``names`` that lives in a bucket arrived in the current backward pass."""
        return all((n not in self.views) or (n in self._seen) for n in names)

    def finish(self) -> None:
        """Buckets that never completed (parameters without a gradient this step) are zero-filled for the missing entries and
        reduced now; then the compute stream is made to wait for every collective."""
        if self._finished:
            return
        for b, idxs in enumerate(self.buckets):
            if b not in self._launched:
                for i in idxs:
                    if self.names[i] not in self._ever:
                        self.views[self.names[i]].zero_()
                self._missing[b] = 0
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        if self._xs is not None:
            torch.cuda.current_stream(self.flat[0].device).wait_stream(self._xs)
        self._finished = True                                      # the next deliver() starts a fresh step
        for i, p in enumerate(self.params):                        # autograd may have replaced .grad: point it back at the bucket
            if p.grad is None or p.grad.data_ptr() != self.views[self.names[i]].data_ptr():
                p.grad = self.views[self.names[i]]
