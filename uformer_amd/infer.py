"""Arbitrary-resolution inference.

``restore``: exactly the steps the reference's evaluation scripts put around ``Uformer.forward`` -- ``expand2square`` to a square
multiple of 128 (test/test_sidd.py:79-92), forward, ``masked_select`` crop, clamp (:106-109; test/test_gopro_hide.py:77-103) -- with
the pad and the crop + clamp each ONE kernel on the device (``uf_expand2square``, ``uf_crop_clamp``); parity-exact with the
reference pipeline.

``restore_tiled``: a capability the reference does not have.  A 1280x720 frame padded to 1280x1280 computes 1.78x the pixels it
needs; here the frame is covered by overlapping SQUARE tiles (the model needs H == W, multiples of 128) that are restored
independently -- as one batch -- and cross-faded over the overlap.  It is an approximation of the full-frame result (a tile's
border pixels see zero padding / a cut context instead of the neighbouring image), which is why it is opt-in and why the overlap is
wide; ``restore`` stays the reference-exact path.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def expand2square(timg: Tensor, factor: float = 128.0) -> Tuple[Tensor, Tensor]:
    """(B,C,h,w) -> zero canvas (B,C,X,X) with the image centred at ((X-h)//2, (X-w)//2), X = max(h,w) rounded up to a
    multiple of ``factor`` (128 = 4 down-samplings x window 8), and the (B,1,X,X) mask of ones over the image: the reference
    helper (written for B = 1) for a batch, one kernel."""
    return ops.expand2square(timg, factor, with_mask=True)


def crop_to_mask(restored: Tensor, h: int, w: int, clamp: bool = False) -> Tensor:
    """Inverse of expand2square for the restored canvas: the (B,C,h,w) region the mask covers
    (``torch.masked_select(restored, mask.bool()).reshape(1,3,h,w)`` in the reference), optionally clamped to [0,1]."""
    return ops.crop_clamp(restored, h, w, clamp)


@torch.no_grad()
def restore(model, img: Tensor, factor: float = 128.0, clamp: bool = True) -> Tensor:
    """Restore images of any (h, w): pad to a square multiple of ``factor``, run ``model``, crop back, clamp to [0,1]
    (test/test_sidd.py:106-109).  ``img``: (B,3,h,w) float32 on the model's device."""
    if img.dim() != 4:
        raise ValueError(f"restore expects (B,C,h,w), got {tuple(img.shape)}")
    h, w = img.shape[-2:]
    padded, _ = ops.expand2square(img, factor, with_mask=False)
    return ops.crop_clamp(model(padded), h, w, clamp)


def _starts(n: int, tile: int, min_overlap: int) -> List[int]:
    if n <= tile:
        return [0]
    k = max(2, math.ceil((n - min_overlap) / (tile - min_overlap)))
    return sorted({round(i * (n - tile) / (k - 1)) for i in range(k)})


@torch.no_grad()
def restore_tiled(model, img: Tensor, tile: int = 768, min_overlap: int = 128, clamp: bool = True, max_batch: int = 8) -> Tensor:
    """EXPERIMENTAL, beyond the reference (which always pads the whole frame to a square, test/test_sidd.py:106-109): an APPROXIMATION by
    construction -- the network's receptive field spans several hundred pixels, so a tile does not see what the full frame sees.
    The mechanics (cutting, padding, ramps, normalisation) are exact: tests/test_host_logic.py checks them to 2e-6 with a pointwise
    model; against the full-frame result of a real network only a PSNR is reported (tests/test_gpu_tail.py).
    Overlapped-tile restoration of (B,3,h,w) images: square ``tile`` x ``tile`` windows (a multiple of 128) at evenly spaced
    positions with at least ``min_overlap`` pixels in common, forwarded in batches of ``max_batch`` tiles, blended with a linear
    ramp across each overlap.  An image that fits one tile takes the reference-exact ``restore`` path."""
    if tile % 128:
        raise ValueError("tile must be a multiple of 128")
    B, C, h, w = img.shape
    if h <= tile and w <= tile:
        return restore(model, img, 128.0, clamp)
    ys, xs = _starts(h, tile, min_overlap), _starts(w, tile, min_overlap)
    acc = torch.zeros((B, C, h, w), dtype=torch.float32, device=img.device)
    wsum = torch.zeros((1, 1, h, w), dtype=torch.float32, device=img.device)

    def ramp(n: int, start: int, starts: List[int], size: int) -> Tensor:
        """weight of a tile along one axis: 1 inside, linear ramps over the parts shared with the neighbouring tiles"""
        wgt = torch.ones(size, device=img.device)
        i = starts.index(start)
        if i > 0:
            ov = starts[i - 1] + tile - start
            if ov > 0:
                wgt[:ov] = (torch.arange(ov, device=img.device) + 0.5) / ov
        if i + 1 < len(starts):
            ov = start + tile - starts[i + 1]
            if ov > 0:
                wgt[size - ov:size] = wgt[size - ov:size] * (1 - (torch.arange(ov, device=img.device) + 0.5) / ov)
        return wgt

    jobs = [(y, x) for y in ys for x in xs]
    for j0 in range(0, len(jobs), max(1, max_batch // B)):
        chunk = jobs[j0:j0 + max(1, max_batch // B)]
        tiles = []
        for (y, x) in chunk:
            th, tw = min(tile, h - y), min(tile, w - x)
            t = img[:, :, y:y + th, x:x + tw]
            if th < tile or tw < tile:                       # image smaller than a tile along this axis: zero-pad at the far side
                t = torch.nn.functional.pad(t, (0, tile - tw, 0, tile - th))
            tiles.append(t)
        out = model(torch.cat(tiles, 0).contiguous())
        for k, (y, x) in enumerate(chunk):
            th, tw = min(tile, h - y), min(tile, w - x)
            wy, wx = ramp(h, y, ys, th), ramp(w, x, xs, tw)
            wgt = (wy[:, None] * wx[None, :])[None, None]
            acc[:, :, y:y + th, x:x + tw] += out[k * B:(k + 1) * B, :, :th, :tw].float() * wgt
            wsum[:, :, y:y + th, x:x + tw] += wgt
    res = acc / wsum
    return torch.clamp(res, 0, 1) if clamp else res


class GraphedForward:
    """Small-batch serving: ``Uformer.forward`` for ONE input shape captured in a HIP graph and replayed.

    The reference's evaluation scripts restore one image per call (test/test_sidd.py:101-107: batch 1), and a forward is ~90 kernel launches:
    at batch 1-4 the launches and the gaps between them are a large part of the step.  ``GraphedForward(model, example)`` runs the forward once
    eagerly on a private stream (packs the weights, allocates that stream's workspace), captures a second run into a ``torch.cuda.CUDAGraph``
    (the library's own side-stream fork / join is followed by the capture, tests/test_gpu_model.py::test_forward_captured_in_a_hip_graph) and
    ``__call__`` copies the new batch into the static input buffer and replays -- bit-identical to the eager forward.  The returned tensor
    is the graph's static output: clone it if it must survive the next call.  Weights are read at replay time (the graph holds
    pointers into the packed-weight buffers): ``__call__`` compares the model's packed-weight cache (keyed by (storage, version) of every parameter) with the one it
    captured and recaptures by itself when the parameters changed; the captured pack is kept alive by this object, so a stale replay can never read freed memory."""

    def __init__(self, model, example: Tensor):
        if not example.is_cuda:
            raise ValueError("GraphedForward needs a CUDA example input")
        self.model = model
        self._x = example.detach().clone()
        self._stream = torch.cuda.Stream(device=example.device)
        self.recapture()

    @torch.no_grad()
    def recapture(self) -> None:
        st = self._stream
        st.wait_stream(torch.cuda.current_stream(self._x.device))
        with torch.cuda.stream(st):
            self.model(self._x)                               # warm-up on the capture stream: packed weights + this stream's workspace
        torch.cuda.current_stream(self._x.device).wait_stream(st)
        torch.cuda.synchronize(self._x.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=st):
            self._y = self.model(self._x)
        # the graph holds raw pointers into THIS pack: keep it alive (an eager forward after a parameter change repacks and would free it under the
        # graph), and remember which pack it is so that __call__ notices the change
        self._captured_pack = self._current_pack()

    def _current_pack(self):
        get = getattr(self.model, "_get_packed", None)
        return get(self._x.device) if get is not None else None

    @torch.no_grad()
    def __call__(self, x: Tensor) -> Tensor:
        if x.shape != self._x.shape or x.dtype != self._x.dtype or x.device != self._x.device:
            raise ValueError(f"GraphedForward was captured for {tuple(self._x.shape)} {self._x.dtype} on {self._x.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        if self._current_pack() is not self._captured_pack:
            # parameters changed since the capture (optimizer step, load_state_dict, .to()): _get_packed has just repacked them into NEW buffers, the
            # graph would replay on the old ones and return stale results -- capture again
            self.recapture()
        self._x.copy_(x)
        self._graph.replay()
        return self._y


class PipelinedForward:
    """Throughput serving, beyond the reference (whose scripts restore one batch after the other, test/test_sidd.py:101-107): up to ``depth``
    forwards of successive batches IN FLIGHT at once, each on a stream of its own.

    One forward is a chain of ~90 dependent launches, and the deep stages of a 256 x 256 batch do not fill the chip (Uformer-B at batch 16: 64
    windows in the bottleneck, 256 = one per CU at 32 x 32): with one batch in flight those launches run at the latency of a single window while
    most of the matrix pipes idle.  Batches are independent, so the next batch's high-resolution stages can run beside them.  ``submit(x)``
    enqueues a forward on the next stream of a small ring (ordered behind everything the caller's stream has enqueued so far -- the producer of
    ``x``) and returns a handle; ``handle.result()`` makes the CALLER's stream wait for that forward and returns its output (every handle must be
    consumed).  Results are the
    eager forward's bit for bit (the kernels are batch-size and stream invariant, tests/test_gpu_model.py).  The caller keeps ``x`` alive until
    ``result()``; outputs are fresh tensors.  Parameters must not change while forwards are in flight (the small f32 parameters are read in place,
    and an update enqueued on the caller's stream is not ordered behind the ring streams): consume every handle, or call ``drain()``, first.
    Measured on an MI355X, Uformer-B 256 x 256 bf16 (profiles/r06_run21_pipelined.txt): batch 16: 2419 img/s eager, 2566 with two forwards in flight;
    batch 4: 1190 -> 1836 (2) -> 2177 (3); batch 1: 389 -> 696 (2) -> 945 img/s (3)."""

    class Handle:
        def __init__(self, y: Tensor, ev, stream):
            self._y, self._ev, self._stream = y, ev, stream

        def result(self) -> Tensor:
            cur = torch.cuda.current_stream(self._y.device)
            cur.wait_event(self._ev)
            self._y.record_stream(cur)          # allocated on the ring stream, consumed on the caller's
            return self._y

    def __init__(self, model, depth: int = 2, device=None):
        if depth < 1:
            raise ValueError("depth must be at least 1")
        dev = device if device is not None else next(model.parameters()).device
        self.model, self.depth = model, depth
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self._k = 0

    @torch.no_grad()
    def submit(self, x: Tensor) -> "PipelinedForward.Handle":
        s = self._streams[self._k % self.depth]
        self._k += 1
        # the operand pack is (re)built HERE, on the caller's stream, which every ring stream waits for -- not inside the forward on one ring stream while
        # the others already read it; the handle keeps the pack it ran on alive (a later repack frees the old buffers only after result())
        get = getattr(self.model, "_get_packed", None)
        pack = get(x.device) if get is not None else None
        s.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(s):
            y = self.model(x)
            ev = torch.cuda.Event()
            ev.record(s)
        x.record_stream(s)
        h = PipelinedForward.Handle(y, ev, s)
        h._pack = pack
        return h

    def drain(self) -> None:
        """the caller's stream waits for every forward submitted so far (before a parameter update, a checkpoint load, ...)"""
        cur = torch.cuda.current_stream(self._streams[0].device)
        for s in self._streams:
            cur.wait_stream(s)

    @torch.no_grad()
    def map(self, batches):
        """Generator: forwards of an iterable of batches, ``depth`` in flight, outputs in order."""
        pending = []
        for x in batches:
            pending.append(self.submit(x))
            if len(pending) >= self.depth:
                yield pending.pop(0).result()
        while pending:
            yield pending.pop(0).result()
