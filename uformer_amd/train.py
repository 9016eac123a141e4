"""Forward + backward of the whole path for training, assembled from the C-ABI kernels.

Forward: every LeWin block runs on the two FUSED inference kernels (``uf_lewin_block_train_fwd``: attn_block + leff2 with timm's
DropPath scales folded into the residual adds) and keeps ONE tensor per block -- its f32 input (127 MB per image over the 40
blocks of Uformer-B).  Backward: a block recomputes its intermediates from that input with the op-level kernels and then runs the
op-level backward (LayerNorm, projections and their weight gradients, window attention, depthwise stencil both ways, GELU').
Everything that used to be ATen glue is a kernel: bias-table gradient (``uf_rpb_table_grad``), modulator gradient
(``uf_rows_sum``), patch matrices of the strided convolutions (``uf_im2col`` / ``uf_col2im``), Charbonnier loss and AdamW
(uformer_amd/losses.py, optim.py).  PyTorch only permutes layouts (head merge, window order) and adds residuals.  Weight casts /
transposes are made once per step (``BlockPack``).  ``compute_dtype=float32`` at shapes the fused kernels do not cover keeps the
op-by-op forward (it stores the intermediates instead of recomputing them).

model.py:908-989; train/train_denoise.py:180-184.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops, packing

Tensor = torch.Tensor


Saved = Dict[str, object]
Grads = Dict[str, Tensor]


_ZERO_BIAS: Dict[Tuple[int, str], Tensor] = {}


def _zeros(n: int, dev) -> Tensor:
    """a zero bias vector of length n (cached: the input-gradient GEMMs asked for a fresh one 2 300 times per step)"""
    key = (n, str(dev))
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros(n, device=dev)
    return z


def _input_grad(dy: Tensor, w_t: Tensor) -> Tensor:
    """dX = dY W for y = x W^T + b: the forward GEMM with the TRANSPOSED weight ``w_t`` (K, N) and a zero bias."""
    return ops.linear(dy, w_t, _zeros(w_t.shape[0], dy.device))


class BlockPack:
    """Per-step operand copies of one block's parameters: T-typed weights, their transposes for the input gradients, the dense
    bias, the tap tables, and the fragment-major pack the fused forward kernels read."""

    def __init__(self, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, T: torch.dtype, fused: bool):
        f = lambda k: p[prefix + k]                                             # noqa: E731
        self.prefix, self.heads, self.shift, self.T = prefix, heads, shift, T
        self.mod = f("modulator.weight") if (prefix + "modulator.weight") in p else None
        self.wqkv = torch.cat([f("attn.qkv.to_q.weight"), f("attn.qkv.to_kv.weight")], 0).to(T)
        self.bqkv = torch.cat([f("attn.qkv.to_q.bias"), f("attn.qkv.to_kv.bias")], 0)
        self.wp, self.w1, self.w2 = f("attn.proj.weight").to(T), f("mlp.linear1.0.weight").to(T), f("mlp.linear2.0.weight").to(T)
        self.wqkv_t, self.wp_t, self.w1_t, self.w2_t = (w.t().contiguous() for w in (self.wqkv, self.wp, self.w1, self.w2))
        self.w9 = packing.pack_dwconv(f("mlp.dwconv.0.weight"))
        if T in (torch.bfloat16, torch.float16):
            self.w9 = self.w9.to(T).float()          # taps rounded to the operand type, as the fused forward's matrix-pipe stencil and uf_pack_block_train do (round 6)
        self.w9_flip = self.w9.flip(0).contiguous()
        self.bias = packing.rpb_dense(f("attn.relative_position_bias_table"), f("attn.relative_position_index"))
        self.p = p
        self.fused = None
        if fused:
            self.fused, self._keep = packing.pack_block(p, prefix, heads, shift, T)
        self._train_params = None

    @property
    def train_params(self):
        """``uf_block_train_params`` over this pack's tensors (they stay alive with the pack)."""
        if self._train_params is None:
            from . import _lib
            f = lambda k: self.p[self.prefix + k].detach().float().contiguous()      # noqa: E731
            keep = self._tp_keep = dict(norm1_w=f("norm1.weight"), norm1_b=f("norm1.bias"), norm2_w=f("norm2.weight"), norm2_b=f("norm2.bias"),
                                        modulator=None if self.mod is None else self.mod.detach().float().contiguous(), rpb_dense=self.bias.float().contiguous(),
                                        wqkv=self.wqkv.contiguous(), wqkv_t=self.wqkv_t, bqkv=self.bqkv.detach().float().contiguous(), wproj=self.wp.contiguous(),
                                        wproj_t=self.wp_t, bproj=f("attn.proj.bias"), w1=self.w1.contiguous(), w1_t=self.w1_t, b1=f("mlp.linear1.0.bias"),
                                        wdw9=self.w9.float().contiguous(), wdw9_flip=self.w9_flip.float().contiguous(), bdw=f("mlp.dwconv.0.bias"), w2_t=self.w2_t)
            tp = _lib.BlockTrainParams()
            for k, v in keep.items():
                setattr(tp, k, None if v is None else v.data_ptr())
            tp.shift, tp.heads = self.shift, self.heads
            keep["w2"] = self.w2.contiguous()
            tp.w2 = keep["w2"].data_ptr()
            self._train_params = tp
        return self._train_params


_STD_INDEX: Dict[Tuple[int, int], bool] = {}


def _index_is_standard(idx: Tensor) -> bool:
    """relative_position_index equals the reference's (model.py:471-481)?  Checked once per buffer (it is a constant)."""
    key = (idx.data_ptr(), idx._version)
    ok = _STD_INDEX.get(key)
    if ok is None:
        from .spec import relative_position_index
        ok = _STD_INDEX[key] = bool(idx.shape == (64, 64) and torch.equal(idx.detach().cpu().long(), relative_position_index(8)))
    return ok


class NativeBlockPack:
    """The per-step operand pack of one block made by ``uf_pack_block_train`` (5 launches into one buffer): ``fused`` is the
    ``uf_block_params`` of the fused forward, ``train_params`` the ``uf_block_train_params`` of the block-level backward.  The small
    f32 parameters are referenced in place: the pack is valid until the parameters change (the optimizer step)."""

    def __init__(self, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, T: torch.dtype):
        from . import _lib
        self.prefix, self.heads, self.shift, self.T = prefix, heads, shift, T
        f = lambda k: p[prefix + k].detach()                                   # noqa: E731
        names = dict(norm1_w="norm1.weight", norm1_b="norm1.bias", norm2_w="norm2.weight", norm2_b="norm2.bias", rpb_table="attn.relative_position_bias_table",
                     rpb_index="attn.relative_position_index", to_q_w="attn.qkv.to_q.weight", to_q_b="attn.qkv.to_q.bias", to_kv_w="attn.qkv.to_kv.weight",
                     to_kv_b="attn.qkv.to_kv.bias", proj_w="attn.proj.weight", proj_b="attn.proj.bias", lin1_w="mlp.linear1.0.weight", lin1_b="mlp.linear1.0.bias",
                     dw_w="mlp.dwconv.0.weight", dw_b="mlp.dwconv.0.bias", lin2_w="mlp.linear2.0.weight", lin2_b="mlp.linear2.0.bias")
        keep = self._keep = {}
        raw = _lib.BlockRawParams()
        for field, key in names.items():
            t = f(key)
            t = t.contiguous() if field == "rpb_index" else t.float().contiguous()
            if field == "rpb_index" and t.dtype != torch.int64:
                t = t.long()
            keep[field] = t
            setattr(raw, field, t.data_ptr())
        if (prefix + "modulator.weight") in p:
            keep["modulator"] = f("modulator.weight").float().contiguous()
            raw.modulator = keep["modulator"].data_ptr()
        raw.index_is_standard = int(_index_is_standard(keep["rpb_index"]))
        C = keep["norm1_w"].numel()
        dev = keep["norm1_w"].device
        lib = _lib.load()
        dt = ops.uf_dtype(T)
        nbytes = lib.uf_pack_block_train_bytes(C, heads, dt)
        self._buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.fused, self.train_params = _lib.BlockParams(), _lib.BlockTrainParams()
        import ctypes
        with torch.cuda.device(dev):
            _lib.check(lib.uf_pack_block_train(ctypes.byref(raw), C, heads, shift, dt, self._buf.data_ptr(), nbytes, ctypes.byref(self.fused),
                                               ctypes.byref(self.train_params), torch.cuda.current_stream().cuda_stream), "uf_pack_block_train")
        # the same operands as tensors (views into the pack buffer) for the op-by-op forward / backward: the BlockPack attributes
        tp, esz = self.train_params, torch.empty(0, dtype=T).element_size()

        def view(ptr, shape, dtype=T, size=esz):
            off, n = ptr - self._buf.data_ptr(), size
            for d in shape:
                n *= d
            return self._buf[off:off + n].view(dtype).reshape(shape)

        f32 = lambda ptr, shape: view(ptr, shape, torch.float32, 4)            # noqa: E731
        self.wqkv, self.wqkv_t, self.bqkv = view(tp.wqkv, (3 * C, C)), view(tp.wqkv_t, (C, 3 * C)), f32(tp.bqkv, (3 * C,))
        self.wp, self.wp_t = view(tp.wproj, (C, C)), view(tp.wproj_t, (C, C))
        self.w1, self.w1_t, self.w2, self.w2_t = view(tp.w1, (4 * C, C)), view(tp.w1_t, (C, 4 * C)), view(tp.w2, (C, 4 * C)), view(tp.w2_t, (4 * C, C))
        self.w9, self.w9_flip, self.bias = f32(tp.wdw9, (9, 4 * C)), f32(tp.wdw9_flip, (9, 4 * C)), f32(tp.rpb_dense, (heads, 64, 64))
        self.mod = p[prefix + "modulator.weight"] if (prefix + "modulator.weight") in p else None


# ------------------------------------------------------------------------------------------------------------------
# LeWin block (model.py:908-989)
# ------------------------------------------------------------------------------------------------------------------
def lewin_block_forward(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dtype: torch.dtype,
                        drop: Optional[Tensor] = None, pk: Optional[BlockPack] = None, need_y: bool = True) -> Tuple[Optional[Tensor], Saved]:
    """x: (B, L, C) f32 on the GPU -> (y, saved).  Op-by-op forward that keeps what the backward reads (also the RECOMPUTATION a
    block runs at the start of its backward).  ``drop``: None (eval) or (2, B) per-sample DropPath scales bernoulli(keep)/keep of
    the two residual branches (model.py:986-987).  ``need_y=False``: the recomputation stops before linear2 (nothing reads y)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    M = B * L
    T = dtype
    pk = pk or BlockPack(p, prefix, heads, shift, T, fused=fused_attn_covers(T, C, heads))
    f = lambda k: p[prefix + k]                                             # noqa: E731
    x2 = x.reshape(M, C).float().contiguous()
    s1 = drop[0].float().contiguous() if drop is not None else None         # per-sample DropPath scales (B,)
    s2 = drop[1].float().contiguous() if drop is not None else None
    fused_attn = fused_attn_covers(T, C, heads) and getattr(pk, "fused", None) is not None
    if fused_attn:
        # round 6: LN1 -> q/k/v -> attention -> proj + residual -> LN2 -> linear1 in ONE launch (the fused window kernel of the inference path) with side
        # stores of every operand the backward reads, instead of the six launches below and their round trips through HBM
        x1, xn, q, k, vt, o, z, a1 = ops.lewin_attn_train_fwd(pk.fused, x2, B, H, W, heads, T, s1)
    else:
        xn = ops.layernorm(x2, f("norm1.weight"), f("norm1.bias"), B=B, H=H, W=W, dtype=T, windowed=True, shift=shift, modulator=pk.mod)
        q, k, vt = ops.qkv(xn, pk.wqkv, pk.bqkv, heads)                          # window rows; q already scaled
        o = ops.window_attention_core(q, k, vt, pk.bias, H=H, W=W, shift=shift)  # (M, C) window rows
        # x + DropPath(window_reverse(proj(o))): the residual add, the scale and the un-partition in the projection GEMM's store   model.py:975-986
        x1 = ops.linear_residual(o, pk.wp, f("attn.proj.bias"), x2, s1, B, H, W, windowed=True, shift=shift)
        z = ops.layernorm(x1, f("norm2.weight"), f("norm2.bias"), B=B, H=H, W=W, dtype=T)
    if fused_attn or _GELU_IN:     # linear1 keeps only its pre-activation (the backward needs that one); the stencil activates it as it loads it
        if not fused_attn:
            a1 = ops.linear(z, pk.w1, f("mlp.linear1.0.bias"))
        h1 = None
        c, g2 = ops.dwconv3x3_pre_gelu(a1.reshape(B, H, W, 4 * C), pk.w9, f("mlp.dwconv.0.bias"), gelu_in=True)
    else:
        a1, h1 = ops.linear_pre_gelu(z, pk.w1, f("mlp.linear1.0.bias"))      # pre-activation (kept for GELU') and activation, one pass
        h1 = h1.reshape(B, H, W, 4 * C)
        c, g2 = ops.dwconv3x3_pre_gelu(h1, pk.w9, f("mlp.dwconv.0.bias"))    # likewise for the stencil and the second GELU
    g2 = g2.reshape(M, 4 * C)
    y = None
    if need_y:
        y = ops.linear_residual(g2, pk.w2, f("mlp.linear2.0.bias"), x1, s2, B, H, W).reshape(B, L, C)     # x1 + DropPath(linear2(.))   model.py:987
    saved = dict(s1=s1, s2=s2, p=p, prefix=prefix, heads=heads, shift=shift, T=T, shape=(B, L, C), x2=x2, xn=xn, q=q, k=k, vt=vt, o=o, x1=x1, z=z, a1=a1,
                 h1=h1, c=c, g2=g2, pk=pk, mod=pk.mod is not None)
    return y, saved


_SIDE_STREAMS: Dict[str, list] = {}
_FUSED_ATTN_FWD = True     # False: the op-by-op attention half of the kept-intermediates forward (tests compare the two)


def fused_attn_covers(T: torch.dtype, C: int, heads: int) -> bool:
    """shapes uf_lewin_attn_train_fwd is built for: 2-byte operands, head_dim 32, C = 32 ... 512"""
    return _FUSED_ATTN_FWD and T in (torch.bfloat16, torch.float16) and C == 32 * heads and C in (32, 64, 128, 256, 512)


# module attributes, not environment switches (round 5): the shipped forms; the alternatives stay reachable for tests / A-B runs by setting the attribute
_GELU_IN = True            # False: linear1 writes pre-activation AND activation, the stencil reads the latter
_DW_BWD_FUSED = True       # False: the two-kernel depthwise backward (uf_dwconv3x3_mul_dgelu + uf_dwconv3x3_wgrad)
_VERBOSE = False           # True: say which training form (kept-intermediates / recompute) was chosen and why
_NATIVE_PACK = True        # False: per-step operand packing through ATen (BlockPack) instead of uf_pack_block_train


class _Side:
    """Weight-gradient jobs of a block on a side stream (nothing downstream in the block reads them), as the C++ block backward does
    (UF_BWD_STREAMS=1 turns it off): ``run(fn)`` orders the side stream behind everything enqueued on the caller's stream so far and runs
    ``fn`` on it; ``join()`` makes the caller's stream wait for the side stream.  Allocations made inside ``run`` belong to the side
    stream; inputs stay referenced by the caller until after ``join()``, so the caching allocator cannot hand them out early."""

    def __init__(self, device):
        n = int(os.environ.get("UF_BWD_STREAMS", "2"))           # 1: off;  2: one side stream;  3+: jobs alternate over n - 1 side streams
        self.on = n > 1
        if self.on:
            key = str(device)
            have = _SIDE_STREAMS.setdefault(key, [])
            while len(have) < n - 1:
                have.append(torch.cuda.Stream(device=device))
            self.sides, self.cur, self.k = have[:n - 1], torch.cuda.current_stream(device), 0

    def run(self, fn):
        if not self.on:
            return fn()
        side = self.sides[self.k % len(self.sides)]
        self.k += 1
        side.wait_stream(self.cur)
        with torch.cuda.stream(side):
            return fn()

    def join(self):
        if self.on:
            for side in self.sides:
                self.cur.wait_stream(side)


_FUSE_FORK = True          # False: separate grad_fork passes (tests set the attribute)
_NO_CAST = object()


def lewin_block_backward(sv: Saved, dy: Tensor, dyT: Optional[Tensor] = None, next_scale=_NO_CAST):
    """dy: (B, L, C) gradient of the block output -> (dx, gradients keyed like the reference's named_parameters()).
    ``dyT``: T(dy * s2) if the caller already has it (the LayerNorm backward of the block that ran AFTER this one wrote it next to its dx).
    ``next_scale`` (the LeFF DropPath scales (B,) of the block that ran BEFORE this one, or None for no scaling): also return
    T(dx * next_scale) as a third result -- that block's ``dyT``."""
    p, prefix, heads, shift, T = sv["p"], sv["prefix"], sv["heads"], sv["shift"], sv["T"]
    side = _Side(dy.device)
    B, L, C = sv["shape"]
    H = W = int(math.sqrt(L))
    M, hd = B * L, C // heads
    f = lambda k: p[prefix + k]                                             # noqa: E731
    g: Grads = {}
    dyf = dy.reshape(M, C).float()
    if dyT is None:
        _, dyT = ops.grad_fork(dyf, None, sv["s2"], B, H, W, T)              # gradient entering the (scaled) LeFF branch, as a GEMM operand
    # LeFF: linear2 -> GELU -> depthwise -> GELU -> linear1                                   (model.py:666-685)
    g[prefix + "mlp.linear2.0.weight"], g[prefix + "mlp.linear2.0.bias"] = side.run(lambda: ops.linear_wgrad(dyT, sv["g2"]))
    pk: BlockPack = sv["pk"]
    dc = ops.linear_mul_dgelu(dyT, pk.w2_t, _zeros(4 * C, dyT.device), sv["c"].reshape(M, 4 * C)).reshape(B, H, W, 4 * C)   # dY W2, times GELU'(c)
    if _DW_BWD_FUSED:      # flipped-tap stencil times GELU'(a1) AND the tap / bias gradients, one pass over dc (h1 recomputed from a1)
        da1, dw9, g[prefix + "mlp.dwconv.0.bias"] = ops.dwconv3x3_bwd(dc, pk.w9_flip, sv["a1"].reshape(B, H, W, 4 * C))
        da1 = da1.reshape(M, 4 * C)
        g[prefix + "mlp.dwconv.0.weight"] = dw9.t().reshape(4 * C, 1, 3, 3)
    else:
        def _dw():
            h1 = sv["h1"] if sv["h1"] is not None else ops.gelu(sv["a1"]).reshape(B, H, W, 4 * C)
            dw9, db = ops.dwconv3x3_wgrad(h1, dc)
            return dw9.t().reshape(4 * C, 1, 3, 3), db
        g[prefix + "mlp.dwconv.0.weight"], g[prefix + "mlp.dwconv.0.bias"] = side.run(_dw)
        da1 = ops.dwconv3x3_mul_dgelu(dc, pk.w9_flip, sv["a1"].reshape(B, H, W, 4 * C)).reshape(M, 4 * C)   # flipped-tap stencil, times GELU'(a1)
    g[prefix + "mlp.linear1.0.weight"], g[prefix + "mlp.linear1.0.bias"] = side.run(lambda: ops.linear_wgrad(da1, sv["z"]))
    dz = _input_grad(da1, pk.w1_t)
    # attention half: proj -> attention -> qkv -> (+modulator) -> partition/roll -> LN1              (model.py:951-986)
    # dx1 = LN2-path gradient + dy (the residual), and the (scaled) gradient entering the attention branch in window order
    if _FUSE_FORK and dz.dtype == T:                                        # both from the LayerNorm backward kernel
        dx1, g[prefix + "norm2.weight"], g[prefix + "norm2.bias"], dyw = ops.layernorm_bwd_fused(sv["x1"], f("norm2.weight"), dz, B, H, W, add=dyf,
                                                                                                 cast=dict(scale=sv["s1"], windowed=True, shift=shift))
    else:
        dx1, g[prefix + "norm2.weight"], g[prefix + "norm2.bias"] = ops.layernorm_bwd_fused(sv["x1"], f("norm2.weight"), dz, B, H, W)
        dx1, dyw = ops.grad_fork(dx1, dyf, sv["s1"], B, H, W, T, windowed=True, shift=shift, want_sum=True)
    g[prefix + "attn.proj.weight"], g[prefix + "attn.proj.bias"] = side.run(lambda: ops.linear_wgrad(dyw, sv["o"]))
    do = _input_grad(dyw, pk.wp_t)
    dqkv, dbias = ops.window_attention_bwd_qkv(sv["q"], sv["k"], sv["vt"], pk.bias, do, H, W, shift)    # heads merged, dq times the query scale
    g[prefix + "attn.relative_position_bias_table"] = side.run(lambda: ops.rpb_table_grad(dbias))   # gather over the pairs of each table entry: deterministic
    nW = M // 64
    dWqkv, dbqkv = side.run(lambda: ops.linear_wgrad(dqkv, sv["xn"]))
    g[prefix + "attn.qkv.to_q.weight"], g[prefix + "attn.qkv.to_kv.weight"] = dWqkv[:C], dWqkv[C:]
    g[prefix + "attn.qkv.to_q.bias"], g[prefix + "attn.qkv.to_kv.bias"] = dbqkv[:C], dbqkv[C:]
    dxn = _input_grad(dqkv, pk.wqkv_t)
    if sv["mod"]:                                                           # the (64, C) table is added to every window
        g[prefix + "modulator.weight"] = side.run(lambda: ops.rows_sum(dxn.reshape(nW, 64 * C)).reshape(64, C))
    # LN1 backward reads dxn in window order (window_reverse + roll back folded in) and adds the residual path's gradient
    if next_scale is not _NO_CAST and _FUSE_FORK and dxn.dtype == T:
        dx, g[prefix + "norm1.weight"], g[prefix + "norm1.bias"], dyT_next = ops.layernorm_bwd_fused(sv["x2"], f("norm1.weight"), dxn, B, H, W, add=dx1, windowed=True,
                                                                                                     shift=shift, cast=dict(scale=next_scale, windowed=False))
        side.join()
        return dx.reshape(B, L, C), g, dyT_next
    dx, g[prefix + "norm1.weight"], g[prefix + "norm1.bias"] = ops.layernorm_bwd_fused(sv["x2"], f("norm1.weight"), dxn, B, H, W, add=dx1, windowed=True, shift=shift)
    side.join()
    if next_scale is not _NO_CAST:
        return dx.reshape(B, L, C), g, None
    return dx.reshape(B, L, C), g


def _named_block_grads(prefix: str, gv: Dict[str, Tensor], C: int) -> Grads:
    """uf_block_grads fields -> the reference's parameter names (the fused q|k|v gradient splits into to_q and to_kv)."""
    g = {prefix + "norm1.weight": gv["norm1_w"], prefix + "norm1.bias": gv["norm1_b"], prefix + "norm2.weight": gv["norm2_w"], prefix + "norm2.bias": gv["norm2_b"],
         prefix + "attn.relative_position_bias_table": gv["rpb_table"],
         prefix + "attn.qkv.to_q.weight": gv["wqkv"][:C], prefix + "attn.qkv.to_kv.weight": gv["wqkv"][C:],
         prefix + "attn.qkv.to_q.bias": gv["bqkv"][:C], prefix + "attn.qkv.to_kv.bias": gv["bqkv"][C:],
         prefix + "attn.proj.weight": gv["wproj"], prefix + "attn.proj.bias": gv["bproj"],
         prefix + "mlp.linear1.0.weight": gv["w1"], prefix + "mlp.linear1.0.bias": gv["b1"],
         prefix + "mlp.dwconv.0.weight": gv["wdw"], prefix + "mlp.dwconv.0.bias": gv["bdw"],
         prefix + "mlp.linear2.0.weight": gv["w2"], prefix + "mlp.linear2.0.bias": gv["b2"]}
    if "modulator" in gv:
        g[prefix + "modulator.weight"] = gv["modulator"]
    return g


def lewin_block_forward_backward(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dy: Tensor,
                                 dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Grads]:
    y, sv = lewin_block_forward(x, p, prefix, heads, shift, dtype)
    dx, g = lewin_block_backward(sv, dy)
    return y, dx, g


# ------------------------------------------------------------------------------------------------------------------
# convolutions of the samplers / stem / head as patch GEMMs: dW = dY^T cols, dX = fold(dY W)   (4 % of the FLOPs)
# ------------------------------------------------------------------------------------------------------------------
def _pad_rows8(t: Tensor) -> Tensor:
    n = t.shape[0]
    return t if n % 8 == 0 else F.pad(t, (0, 0, 0, 8 - n % 8))


def _conv_backward(x_src: Tensor, nchw: bool, geom: Tuple[int, int, int, int], w: Tensor, dy_rows: Tensor, stride: int, padding: int, T: torch.dtype,
                   add_to: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Gradients of a Conv2d from its input (f32 token rows (B*H*W, Cin), or the NCHW image when ``nchw``), weight
    (Cout,Cin,k,k) and output gradient rows (B*Ho*Wo, Cout): returns (dx in the layout of the input, dW, db).  The patch matrix
    comes from uf_im2col, both products run through uf_linear_wgrad / uf_linear_fwd, uf_col2im gathers the input gradient
    (``add_to``: accumulated into that tensor, e.g. the skip gradient).  Out-channel counts are padded to 8 (zero columns)."""
    B, H, W, Cin = geom
    Cout, _, k, _ = w.shape
    cols = ops.im2col(x_src, B, H, W, Cin, k, stride, padding, T, nchw=nchw)          # (M_out, ldc)
    ldc, Kc = cols.shape[1], k * k * Cin
    dyT = dy_rows.to(T)
    if Cout % 8:
        dyT = F.pad(dyT, (0, 8 - Cout % 8))
    dyT = dyT.contiguous()
    wmat = F.pad(w.permute(0, 2, 3, 1).reshape(Cout, Kc), (0, ldc - Kc, 0, dyT.shape[1] - Cout)).to(T)    # (Cout padded, ldc), column (ky,kx,c)
    dWm, db = ops.linear_wgrad(dyT, cols)
    dcols = _input_grad(dyT, wmat.t().contiguous())                                    # (M_out, ldc)
    dx = ops.col2im(dcols, B, H, W, Cin, k, stride, padding, nchw=nchw, out=add_to)
    dW = dWm[:Cout, :Kc].reshape(Cout, k, k, Cin).permute(0, 3, 1, 2).contiguous()
    return dx, dW, db[:Cout]


def _conv3x3_backward(x: Tensor, nchw: bool, geom: Tuple[int, int, int, int], w: Tensor, dy_rows: Tensor, T: torch.dtype, act_out: Optional[Tensor] = None,
                      need_dx: bool = True) -> Tuple[Optional[Tensor], Tensor, Tensor]:
    """Backward of InputProj / OutputProj (3x3, stride 1, pad 1; LeakyReLU' folded in when ``act_out`` is given): the direct f32
    kernels for the reference's shapes (embed_dim 16 / 32: a side of <= 4 channels), the patch-matrix route for anything else."""
    B, H, W, Cin = geom
    Cout = w.shape[0]
    direct = (nchw and Cin <= 4 and Cout in (16, 32, 64)) or (not nchw and Cout <= 4 and Cin % 4 == 0)
    if direct and Cin * Cout * 9 <= 2048:
        return ops.conv3x3_bwd(x, dy_rows, w, B, H, W, nchw=nchw, act_out=act_out, slope=0.01, need_dx=need_dx)
    if act_out is not None:
        dy_rows = dy_rows * torch.where(act_out > 0, torch.ones_like(act_out), torch.full_like(act_out, 0.01))     # LeakyReLU(0.01), model.py:786
    return _conv_backward(x, nchw, geom, w, dy_rows, 1, 1, T)


_RECOMPUTE_CHOICE: Dict[tuple, bool] = {}


def choose_recompute(cfg, B: int, H: int, device) -> bool:
    """Kept-intermediates or recompute form for (arch, batch, resolution) on this device -- decided ONCE and remembered, so every step of a run (and,
    under DDP, every rank: the choice is reduced with MAX over the process group) takes the same kernels and rounds the same way (the
    per-forward decision compared against the driver's free memory, which excludes what PyTorch's caching allocator reserved on the previous step,
    and could flip between steps or differ between ranks).  The memory that counts as available = the driver's free bytes + the allocator's
    reserved-but-unallocated bytes."""
    dev_ = torch.device(device)
    key = (tuple(cfg.depths), cfg.embed_dim, B, H, dev_.index if dev_.index is not None else torch.cuda.current_device())   # a bare "cuda" names the current device
    # (Under DDP every rank must reach this point with the same (arch, batch, resolution) the first time: the choice is reduced over the process group.)
    if key not in _RECOMPUTE_CHOICE:
        dims, div = cfg.stage_dims(), cfg.stage_res_div()
        per_tc = 52 if _GELU_IN else 60           # bytes per token x channel of a block (measured 49.0 / 56.5 GB for Uformer-B 256^2 at batch 32); round 4: linear1's activation is not kept
        need = per_tc * B * sum(cfg.depths[s] * (H // div[s]) ** 2 * dims[s] for s in range(9))
        free = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        choice = need > min(0.5 * free, 96e9)      # past ~100 GB the two forms measure the same (batch 64: 356 vs 357 img/s): keep the small one
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            flag = torch.tensor([1.0 if choice else 0.0], device=device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            choice = bool(flag.item() > 0)
        _RECOMPUTE_CHOICE[key] = choice
        if _VERBOSE:
            print(f"[uformer_amd.train] batch {B} at {H}x{H}: {'recompute' if choice else 'kept-intermediates'} form ({need / 1e9:.1f} GB of intermediates, {free / 1e9:.1f} GB available)", flush=True)
    return _RECOMPUTE_CHOICE[key]


class UformerTape:
    """One forward of the whole model (model.py:1269-1305) that keeps what the reverse sweep reads, and that sweep.
    ``drop_scales``: None (eval semantics) or a (2 * n_blocks, B) tensor of DropPath scales in execution order.
    ``recompute`` (default: on whenever the fused kernels cover the operand type): a block keeps only its input and rebuilds its
    intermediates at the start of its backward; off: the op-by-op forward keeps them (the round-1 form, ~18x the memory)."""

    def __init__(self, sd: Dict[str, Tensor], cfg, dtype: torch.dtype = torch.float32, drop_scales: Optional[Tensor] = None,
                 recompute: Optional[bool] = None, on_stage_done=None):
        self.sd, self.cfg, self.T, self.drop = sd, cfg, dtype, drop_scales
        if recompute is None and os.environ.get("UF_TRAIN_RECOMPUTE") is not None:       # A/B switch: 0 = keep every intermediate, 1 = always recompute
            recompute = os.environ["UF_TRAIN_RECOMPUTE"] != "0"
        # None = decided in forward() from the batch and the free device memory (2-byte operand types; f32 always keeps its intermediates)
        self.recompute = recompute if (recompute is not None or dtype in (torch.bfloat16, torch.float16)) else False
        self.on_stage_done = on_stage_done          # callback({name: gradient}) for every group of parameters whose gradients are final, in reverse-sweep order

    def forward(self, img: Tensor) -> Tensor:
        from .spec import STAGES
        sd, cfg, T = self.sd, self.cfg, self.T
        B, _, H, W = img.shape
        self.B, self.H = B, H
        if self.recompute is None:
            # Keeping every intermediate of the op-by-op forward costs ~52 bytes per token x channel of every block (measured: 49 GB for
            # Uformer-B 256^2 at batch 32 against 17 GB) and saves the recomputation in the backward: 371 vs 334 img/s on an MI355X
            # (profiles/r03_host.txt).  288 GB of HBM is there to be used: keep them while that is under half of the free memory.
            self.recompute = choose_recompute(cfg, B, H, img.device)
        # widths the training kernels cover, checked once with a clear message (they used to fail deep inside the tape with UF_ERR_SHAPE)
        dims_ = cfg.stage_dims()
        for s_ in range(9):
            hd_ = dims_[s_] // max(1, cfg.num_heads[s_])
            if dims_[s_] % cfg.num_heads[s_] or hd_ not in (16, 32) or dims_[s_] % 16:
                raise ops.UformerHipError(f"training: stage {s_} has {dims_[s_]} channels over {cfg.num_heads[s_]} heads (head_dim {hd_}); supported: head_dim 16 or 32, "
                                          f"channels a multiple of 16 (every get_arch architecture, utils/model_utils.py:56-81)")
        if self.recompute and not any(dims_[s_] == 32 * cfg.num_heads[s_] for s_ in range(9)):
            import warnings
            warnings.warn("the recompute form was selected (use_checkpoint=True, recompute=True, or chosen from the free memory), but no stage has head_dim 32 (the fused kernels the recompute form is built on): every block keeps its "
                          "intermediates (memory ~18x the recompute form)", stacklevel=2)
        shifts = cfg.block_shifts()
        res = self.res = [H, H // 2, H // 4, H // 8, H // 16, H // 8, H // 4, H // 2, H]
        first = [sum(cfg.depths[:s]) for s in range(9)]
        self.saved_blocks: List[List[Saved]] = [[] for _ in range(9)]
        self.packs: Dict[str, BlockPack] = {}

        self.stage_C = [0] * 9

        def stage_fwd(s: int, t: Tensor) -> Tensor:                             # t: (M, C) f32 token rows
            C = self.stage_C[s] = t.shape[1]
            for i in range(cfg.depths[s]):
                bi = first[s] + i
                prefix = f"{STAGES[s]}.blocks.{i}."
                dr = self.drop[2 * bi:2 * bi + 2] if self.drop is not None else None
                # the fused kernels (and the block-level C backward built on them) cover head_dim 32; a head_dim-16 block (Uformer_T,
                # utils/model_utils.py:66-67) takes the op-by-op forward that keeps its intermediates and the op-level backward
                fusable = self.recompute and C == 32 * cfg.num_heads[s]
                # uf_pack_block_train (5 launches) covers C % 32 == 0; its pack also serves the op-by-op form as tensor views
                native = C % 32 == 0 and C % cfg.num_heads[s] == 0 and _NATIVE_PACK
                pk = self.packs[prefix] = (NativeBlockPack(sd, prefix, cfg.num_heads[s], shifts[s][i], T) if native else
                                           BlockPack(sd, prefix, cfg.num_heads[s], shifts[s][i], T, fused=fusable or fused_attn_covers(T, C, cfg.num_heads[s])))
                if fusable:                                                     # fused kernels; the block's input is all that is kept
                    y = ops.lewin_block_train_fwd(pk.fused, t, B, res[s], res[s], T, None if dr is None else dr[0], None if dr is None else dr[1])
                    self.saved_blocks[s].append(dict(x=t, drop=dr, pk=pk))
                    t = y
                else:
                    y, sv = lewin_block_forward(t.reshape(B, res[s] * res[s], C), sd, prefix, cfg.num_heads[s], shifts[s][i], T, dr, pk)
                    self.saved_blocks[s].append(sv)
                    t = y.reshape(-1, C)
            return t

        # the samplers / stem / head run their inference kernels; their inputs are kept
        self.img = img
        t = ops.input_proj(img, packing.pack_input_proj(sd["input_proj.proj.0.weight"]), sd["input_proj.proj.0.bias"])
        self.stem_out = t
        self.skips, self.down_in, self.up_in = [], [], []
        for s in range(4):
            t = stage_fwd(s, t)
            self.skips.append(t)
            self.down_in.append(t)
            wd = packing.pack_downsample(sd[f"dowsample_{s}.conv.0.weight"], T)
            wd_fm = ops.pack_weight_fm(wd) if (T in (torch.bfloat16, torch.float16) and wd.shape[0] % 16 == 0 and wd.shape[1] % 32 == 0) else None     # the LDS-patch form streams it (round 6)
            t = ops.downsample(t, wd, sd[f"dowsample_{s}.conv.0.bias"], B, res[s], res[s], w_fm=wd_fm)
        t = stage_fwd(4, t)
        for k in range(4):
            self.up_in.append(t)
            # torch.cat([up, skip], -1) (model.py:1288) without the concatenation pass: Upsample scatters straight into the first half of the
            # decoder stage's input rows (the inference path's concat buffer, uf_upsample_fwd with ld_o = 2 Cs), the skip is copied into the second
            skip = self.skips[3 - k]
            Cs = skip.shape[1]
            cat = torch.empty((skip.shape[0], 2 * Cs), dtype=torch.float32, device=skip.device)
            ops.upsample(t, packing.pack_upsample(sd[f"upsample_{k}.deconv.0.weight"], T), sd[f"upsample_{k}.deconv.0.bias"], B, res[4 + k], res[4 + k],
                         out=cat, ld_o=2 * Cs)
            cat[:, Cs:].copy_(skip)
            t = stage_fwd(5 + k, cat)
        self.head_in = t
        return ops.output_proj(t, packing.pack_output_proj(sd["output_proj.proj.0.weight"]), sd["output_proj.proj.0.bias"], B, H, W,
                               img if cfg.dd_in == 3 else None)

    def _block_ws(self, s: int) -> Tensor:
        """one workspace for every block backward of the sweep (sized for the largest stage)"""
        if getattr(self, "_bws", None) is None:
            need = max(ops.lewin_block_bwd_workspace_bytes(self.B, r, r, self.stage_C[i], self.cfg.num_heads[i], self.T) for i, r in enumerate(self.res))
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.img.device)
        return self._bws

    def backward(self, dy: Tensor, need_dimg: bool = True) -> Tuple[Optional[Tensor], Grads]:
        from .spec import STAGES
        sd, cfg, T, B, H, res = self.sd, self.cfg, self.T, self.B, self.H, self.res
        g: Grads = {}

        def done(names):
            if self.on_stage_done is not None:
                self.on_stage_done({n: g[n] for n in names})

        def stage_bwd(s: int, d: Tensor, g: Grads) -> Tensor:
            C = d.shape[1]
            d = d.reshape(B, res[s] * res[s], C)
            names = []
            blocks = self.saved_blocks[s]
            dyT = None                                                            # T(d * s2 of the block about to run): from the previous block's LN1 backward
            while blocks:
                sv = blocks.pop()                                                 # frees the block's saved input as the sweep passes it
                if "x2" not in sv:                                                # only the block input was kept: recomputation + backward in one C call
                    pk, dr = sv["pk"], sv["drop"]
                    dxb, gv = ops.lewin_block_bwd(pk.train_params, sv["x"], d.reshape(-1, C), None if dr is None else dr[0], None if dr is None else dr[1],
                                                  B, res[s], res[s], pk.heads, T, ws=self._block_ws(s))
                    d, gb = dxb.reshape(B, res[s] * res[s], C), _named_block_grads(pk.prefix, gv, C)
                elif blocks and "x2" in blocks[-1]:                               # the block that ran before this one reads T(d * its s2): made here
                    d, gb, dyT = lewin_block_backward(sv, d, dyT, next_scale=blocks[-1]["s2"])
                else:
                    d, gb = lewin_block_backward(sv, d, dyT)
                    dyT = None
                del sv
                g.update(gb)
                names.extend(gb.keys())
            done(names)
            return d.reshape(-1, C)

        dy = dy.float()
        dy_rows = dy.permute(0, 2, 3, 1).reshape(B * H * H, 3)
        C8 = self.head_in.shape[1]
        d, g["output_proj.proj.0.weight"], g["output_proj.proj.0.bias"] = _conv3x3_backward(self.head_in, False, (B, H, H, C8), sd["output_proj.proj.0.weight"], dy_rows, T)
        done(["output_proj.proj.0.weight", "output_proj.proj.0.bias"])
        self.head_in = None
        dskip: List[Tensor] = [None] * 4
        for k in reversed(range(4)):
            d = stage_bwd(5 + k, d, g)
            Cs = self.skips[3 - k].shape[1]
            cup = d.shape[1] - Cs
            dskip[3 - k] = d[:, cup:].contiguous()
            # ConvTranspose2d k2 s2 = four independent 1x1 GEMMs over the 2x2 output pixels of every input pixel (uf_upsample_cat_bwd)
            r = res[4 + k]
            w = sd[f"upsample_{k}.deconv.0.weight"]                                # (Cin, Cout, 2, 2)
            wpk_t = packing.pack_upsample(w, T).t().contiguous()                   # (Cin, 4*Cout), n = (dy*2+dx)*Cout + co
            d, dWp, g[f"upsample_{k}.deconv.0.bias"] = ops.upsample_cat_bwd(d, self.up_in[k], wpk_t, B, r, r)
            g[f"upsample_{k}.deconv.0.weight"] = dWp.reshape(2, 2, cup, w.shape[0]).permute(3, 2, 0, 1).contiguous()
            done([f"upsample_{k}.deconv.0.weight", f"upsample_{k}.deconv.0.bias"])
            self.up_in[k] = None
        d = stage_bwd(4, d, g)
        for s in reversed(range(4)):
            Cs = self.down_in[s].shape[1]
            wd = sd[f"dowsample_{s}.conv.0.weight"]                                  # (2C, C, 4, 4)
            dx, dWp, g[f"dowsample_{s}.conv.0.bias"] = ops.downsample_bwd(self.down_in[s], d, packing.pack_downsample(wd, T).t().contiguous(), B, res[s], res[s],
                                                                         add_to=dskip[s])
            g[f"dowsample_{s}.conv.0.weight"] = dWp.reshape(wd.shape[0], 4, 4, Cs).permute(0, 3, 1, 2).contiguous()
            done([f"dowsample_{s}.conv.0.weight", f"dowsample_{s}.conv.0.bias"])
            self.down_in[s] = self.skips[s] = None
            d = stage_bwd(s, dx, g)
        # InputProj: LeakyReLU(0.01)' (model.py:786) from the sign of the stored output, folded into the conv backward
        dimg, g["input_proj.proj.0.weight"], g["input_proj.proj.0.bias"] = _conv3x3_backward(self.img.float(), True, (B, H, H, self.img.shape[1]), sd["input_proj.proj.0.weight"], d, T,
                                                                                             act_out=self.stem_out, need_dx=need_dimg)
        done(["input_proj.proj.0.weight", "input_proj.proj.0.bias"])
        if need_dimg and cfg.dd_in == 3:
            dimg = dimg + dy                                                      # global residual, model.py:1305
        return dimg, g


def uformer_forward_backward(img: Tensor, sd: Dict[str, Tensor], dy: Tensor, *, cfg, dtype: torch.dtype = torch.float32,
                             drop_scales: Optional[Tensor] = None, recompute: Optional[bool] = None, loss_scale: float = 1.0) -> Tuple[Tensor, Tensor, Grads]:
    """Whole-model forward + backward.  img, dy: (B,3,H,W) f32 on the GPU; sd: the reference state_dict on the GPU; cfg:
    uformer_amd.spec.UformerConfig.  Returns (y, d img, parameter gradients keyed like named_parameters()).
    ``loss_scale`` (float16 operands): the reverse sweep runs on ``dy * loss_scale`` and the results are divided by it, what
    torch.cuda.amp.GradScaler does around the reference's backward (train/train_denoise.py:180-184): activation gradients travel as
    f16 operands, and d loss / d y of a mean loss over millions of pixels is below f16's smallest normal number (6.1e-5)."""
    tape = UformerTape(sd, cfg, dtype, drop_scales, recompute)
    y = tape.forward(img)
    dimg, g = tape.backward(dy * loss_scale if loss_scale != 1.0 else dy)
    if loss_scale != 1.0:
        inv = 1.0 / loss_scale
        dimg = dimg * inv
        g = {k: v * inv for k, v in g.items()}
    return y, dimg, g


class UformerFunction(torch.autograd.Function):
    """torch.autograd entry: ``y = UformerFunction.apply(img, cfg, dtype, drop_scales, names, *params)``; backward() hands the
    parameter gradients of the tape to autograd in the order of ``names`` (buffers such as relative_position_index get None)."""

    @staticmethod
    def forward(ctx, img, cfg, dtype, drop_scales, names, *params):
        sink = names.sink if isinstance(names, NamesWithSink) else None
        recompute = names.recompute if isinstance(names, NamesWithSink) else None
        sd = {n: p.detach() for n, p in zip(names, params)}
        tape = UformerTape(sd, cfg, dtype, drop_scales, recompute=recompute, on_stage_done=None if sink is None else sink.deliver)
        y = tape.forward(img.detach().float().contiguous())
        UformerFunction.last_recompute = bool(tape.recompute)      # which form the last forward took (tests, logging)
        ctx.tape, ctx.names, ctx.sink = tape, names, sink
        ctx.img_needs_grad = img.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        dimg, g = ctx.tape.backward(dy.contiguous(), need_dimg=ctx.img_needs_grad)
        ctx.tape = None                                                           # free the saved activations
        if ctx.sink is not None and ctx.sink.delivered([n for n in ctx.names if n in g]):
            # the gradients already sit in the sink's buckets (= param.grad) and are being all-reduced
            grads = tuple(None for _ in ctx.names)
        else:
            grads = tuple(g.get(n) for n in ctx.names)
        return (dimg if ctx.img_needs_grad else None, None, None, None, None) + grads


class LeWinBlockFunction(torch.autograd.Function):
    """torch.autograd entry of ONE LeWin block (the reference's LeWinTransformerBlock is an ordinary differentiable module,
    model.py:908-989): ``y = LeWinBlockFunction.apply(x, prefix_free_names, heads, shift, dtype, drop, *params)`` with x (B, L, C).
    Forward = the op-by-op forward that keeps its intermediates, backward = the op-level backward (lewin_block_forward /
    lewin_block_backward above); ``drop``: None or (2, B) DropPath scales of the two residual branches."""

    @staticmethod
    def forward(ctx, x, names, heads, shift, dtype, drop, *params):
        p = {n: t.detach() for n, t in zip(names, params)}
        y, sv = lewin_block_forward(x.detach().float().contiguous(), p, "", heads, shift, dtype, drop)
        ctx.sv, ctx.names, ctx.x_needs_grad, ctx.x_dtype = sv, names, x.requires_grad, x.dtype
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        dx, g = lewin_block_backward(ctx.sv, dy.float().contiguous())
        ctx.sv = None
        return (dx.to(ctx.x_dtype) if ctx.x_needs_grad else None, None, None, None, None, None) + tuple(g.get(n) for n in ctx.names)


class NamesWithSink(list):
    """parameter names + the gradient sink (uformer_amd.dist.OverlappedGradientAllReduce) the tape delivers to during backward + the
    recompute choice (True: ``use_checkpoint=True`` was given to the constructor, as the reference's torch.utils.checkpoint per block,
    model.py:1056-1057; None: decided from the batch and the free memory)"""
    sink = None
    recompute = None


_KEEP_PROB: Dict[tuple, Tensor] = {}


def sample_drop_scales(rates: Sequence[float], B: int, device, generator: Optional[torch.Generator] = None) -> Tensor:
    """timm DropPath for every block in execution order: two rows (attention branch, LeFF branch) of per-sample scales
    bernoulli(1 - rate) / (1 - rate); rate 0 -> ones.  (model.py:883, :986-987; schedule :1093-1095.)"""
    key = (tuple(float(r) for r in rates), int(B), str(device))
    keep = _KEEP_PROB.get(key)
    if keep is None:     # (2 * blocks, B) keep probabilities, built once: the per-block form was 3 tiny kernels per branch, 240 launches per step
        keep = _KEEP_PROB[key] = (1.0 - torch.tensor(key[0], dtype=torch.float32).clamp(min=0.0)).repeat_interleave(2)[:, None].expand(-1, B).contiguous().to(device)
    return torch.bernoulli(keep, generator=generator) / keep          # rate 0: bernoulli(1) / 1 = 1
