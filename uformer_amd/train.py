"""Backward of a LeWin block assembled from the C-ABI building blocks (first end-to-end slice).

Every FLOP-carrying step is a HIP kernel behind the C ABI (LayerNorm fwd/bwd, the projections and their input / weight
gradients, window attention fwd/bwd, the depthwise stencil in both directions and its tap gradients, GELU'); PyTorch
only permutes layouts (head merge, window order), adds residuals and scatter-adds the 64x64 bias gradient into the
225-row table -- plumbing.  This is the op-by-op form: it keeps the intermediates of the forward instead of recomputing
them inside fused backward kernels (DESIGN.md section 7 has the fused plan); it exists to make the block's gradients
exact and testable against the reference's autograd before the fused kernels are written.

Eval-mode semantics (DropPath = identity), as the gradient fixtures.  model.py:908-989.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import ops, packing

Tensor = torch.Tensor


def lewin_block_forward_backward(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dy: Tensor,
                                 dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """x, dy: (B, L, C) f32 on the GPU; p: the block's parameters (reference names under ``prefix``).
    Returns (y, dx, grads) with grads keyed like the reference's named_parameters()."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    M, hd = B * L, C // heads
    T = dtype
    f = lambda k: p[prefix + k]                                             # noqa: E731
    x2 = x.reshape(M, C).float().contiguous()
    mod = f("modulator.weight") if (prefix + "modulator.weight") in p else None
    wq, wkv = f("attn.qkv.to_q.weight"), f("attn.qkv.to_kv.weight")
    wqkv = torch.cat([wq, wkv], 0).to(T)
    bqkv = torch.cat([f("attn.qkv.to_q.bias"), f("attn.qkv.to_kv.bias")], 0)
    wp, w1, w2 = f("attn.proj.weight").to(T), f("mlp.linear1.0.weight").to(T), f("mlp.linear2.0.weight").to(T)
    w9 = packing.pack_dwconv(f("mlp.dwconv.0.weight"))
    index = f("attn.relative_position_index")
    bias = packing.rpb_dense(f("attn.relative_position_bias_table"), index)
    scale = hd ** -0.5

    # ------------------------------ forward, keeping what the backward reads ------------------------------
    xn = ops.layernorm(x2, f("norm1.weight"), f("norm1.bias"), B=B, H=H, W=W, dtype=T, windowed=True, shift=shift, modulator=mod)
    q, k, vt = ops.qkv(xn, wqkv, bqkv, heads)                                # window rows; q already scaled
    o = ops.window_attention_core(q, k, vt, bias, H=H, W=W, shift=shift)     # (M, C) window rows
    yw = ops.linear(o, wp, f("attn.proj.bias"))
    x1 = x2 + ops.window_reverse(yw.reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C).float()
    z = ops.layernorm(x1, f("norm2.weight"), f("norm2.bias"), B=B, H=H, W=W, dtype=T)
    a1 = ops.linear(z, w1, f("mlp.linear1.0.bias"))                          # pre-activation, kept for GELU'
    h1 = ops.linear(z, w1, f("mlp.linear1.0.bias"), act=1).reshape(B, H, W, 4 * C)
    c = ops.dwconv3x3(h1, w9, f("mlp.dwconv.0.bias"), gelu=False)            # pre-activation of the second GELU
    g2 = ops.dwconv3x3(h1, w9, f("mlp.dwconv.0.bias"), gelu=True).reshape(M, 4 * C)
    y = x1 + ops.linear(g2, w2, f("mlp.linear2.0.bias")).float()

    # ------------------------------ backward ------------------------------------------------------------------
    g: Dict[str, Tensor] = {}
    dyT = dy.reshape(M, C).to(T).contiguous()
    zK = lambda n: torch.zeros(n, device=x.device)                         # noqa: E731  (bias of the transposed GEMMs)
    # LeFF: linear2 -> GELU -> depthwise -> GELU -> linear1                                   (model.py:666-685)
    g[prefix + "mlp.linear2.0.weight"], g[prefix + "mlp.linear2.0.bias"] = ops.linear_wgrad(dyT, g2)
    dg2 = ops.linear(dyT, w2.t().contiguous(), zK(4 * C))
    dc = ops.gelu_bwd(c.reshape(M, 4 * C), dg2).reshape(B, H, W, 4 * C)
    dw9, g[prefix + "mlp.dwconv.0.bias"] = ops.dwconv3x3_wgrad(h1, dc)
    g[prefix + "mlp.dwconv.0.weight"] = dw9.t().reshape(4 * C, 1, 3, 3)
    dh1 = ops.dwconv3x3(dc, w9.flip(0).contiguous(), None, gelu=False)     # input gradient = flipped-tap stencil
    da1 = ops.gelu_bwd(a1, dh1.reshape(M, 4 * C))
    g[prefix + "mlp.linear1.0.weight"], g[prefix + "mlp.linear1.0.bias"] = ops.linear_wgrad(da1, z)
    dz = ops.linear(da1, w1.t().contiguous(), zK(C)).float()
    dx1, g[prefix + "norm2.weight"], g[prefix + "norm2.bias"] = ops.layernorm_bwd(x1, f("norm2.weight"), dz)
    dx1 = dx1 + dy.reshape(M, C).float()
    # attention half: proj -> attention -> qkv -> (+modulator) -> partition/roll -> LN1              (model.py:951-986)
    dyw = ops.window_partition(dx1.reshape(B, H, W, C), 8, shift).reshape(M, C).to(T)
    g[prefix + "attn.proj.weight"], g[prefix + "attn.proj.bias"] = ops.linear_wgrad(dyw, o)
    do = ops.linear(dyw, wp.t().contiguous(), zK(C))
    dq, dk, dvt, dbias = ops.window_attention_bwd(q, k, vt, bias, do, H, W, shift)
    dtab = torch.zeros_like(f("attn.relative_position_bias_table"), dtype=torch.float32)
    dtab.index_add_(0, index.reshape(-1), dbias.permute(1, 2, 0).reshape(64 * 64, heads))
    g[prefix + "attn.relative_position_bias_table"] = dtab
    nW = M // 64
    merge = lambda t: t.reshape(nW, heads, 64, hd).permute(0, 2, 1, 3).reshape(M, C)      # noqa: E731  (nW,h,64,hd) -> rows
    dqkv = torch.cat([merge(dq.float() * scale).to(T), merge(dk), merge(dvt.reshape(nW, heads, hd, 64).transpose(2, 3))], 1).contiguous()
    dWqkv, dbqkv = ops.linear_wgrad(dqkv, xn)
    g[prefix + "attn.qkv.to_q.weight"], g[prefix + "attn.qkv.to_kv.weight"] = dWqkv[:C], dWqkv[C:]
    g[prefix + "attn.qkv.to_q.bias"], g[prefix + "attn.qkv.to_kv.bias"] = dbqkv[:C], dbqkv[C:]
    dxn = ops.linear(dqkv, wqkv.t().contiguous(), zK(C))
    if mod is not None:                                                     # the (64, C) table is added to every window
        g[prefix + "modulator.weight"] = dxn.float().reshape(nW, 64, C).sum(0)
    dln = ops.window_reverse(dxn.float().reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C)
    dx, g[prefix + "norm1.weight"], g[prefix + "norm1.bias"] = ops.layernorm_bwd(x2, f("norm1.weight"), dln)
    return y.reshape(B, L, C), (dx + dx1).reshape(B, L, C), g
