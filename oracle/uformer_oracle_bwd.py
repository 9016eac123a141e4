"""Explicit backward of the hot path -- TEST INFRASTRUCTURE, like uformer_oracle.py.

The reference has no backward code of its own: it relies on torch autograd over model.py.  This file restates that
backward as explicit formulas, op by op, in the order and with the recomputation boundaries the HIP backward kernels of
the next round will use (a LeWin block keeps only its INPUT; everything else is recomputed inside the backward of the
two fused kernels), so that every kernel has a closed-form CPU statement to be checked against.  No autograd is used
here.  Pinned against gradients produced by the reference's own autograd (tests/golden/make_golden_grad.py ->
tests/test_oracle_golden.py): parity is NOT unpinned.

Only tests may import this module (see the header of uformer_oracle.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import uformer_oracle as O

Tensor = torch.Tensor
Grads = Dict[str, Tensor]
WIN = O.WIN


def _acc(g: Grads, key: str, val: Tensor) -> None:
    g[key] = g[key] + val if key in g else val


# ----------------------------------------------------------------------------------------
# elementwise / normalisation pieces
# ----------------------------------------------------------------------------------------
def gelu_erf_grad(x: Tensor) -> Tensor:
    """d/dx [x Phi(x)] = Phi(x) + x phi(x)  (nn.GELU default, model.py:657-660)."""
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def layer_norm_bwd(x: Tensor, w: Tensor, dy: Tensor, eps: float = 1e-5) -> Tuple[Tensor, Tensor, Tensor]:
    """Backward of nn.LayerNorm over the last dim (model.py:881,888).  Returns (dx, dw, db).

    xhat = (x-mu)*rstd;  dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy*w."""
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
    xhat = (x - mu) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))
    red = tuple(range(x.dim() - 1))
    return dx, (dy * xhat).sum(red), dy.sum(red)


def linear_bwd(x: Tensor, w: Tensor, dy: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """y = x W^T + b  ->  dx = dy W,  dW = dy^T x,  db = sum dy   (nn.Linear, model.py:426-427,489,657,661)."""
    x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
    return dy @ w, dy2.t() @ x2, dy2.sum(0)


# ----------------------------------------------------------------------------------------
# window attention (model.py:494-522)
# ----------------------------------------------------------------------------------------
def window_attention_bwd(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, mask: Optional[Tensor],
                         dout: Tensor) -> Tuple[Tensor, Grads]:
    """x: (B_, N, C) window rows (after LN1 / roll / partition / modulator); dout: gradient of the proj output.
    Recomputes q, k, v, P from x (nothing but x is kept from the forward)."""
    B_, N, C = x.shape
    hd = C // heads
    scale = hd ** -0.5
    Wq, Wkv, Wp = p[prefix + "qkv.to_q.weight"], p[prefix + "qkv.to_kv.weight"], p[prefix + "proj.weight"]
    q = (x @ Wq.t() + p[prefix + "qkv.to_q.bias"]).reshape(B_, N, heads, hd).permute(0, 2, 1, 3) * scale
    kv = x @ Wkv.t() + p[prefix + "qkv.to_kv.bias"]
    k = kv[..., :C].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    v = kv[..., C:].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    index = p[prefix + "relative_position_index"]
    s = q @ k.transpose(-2, -1) + O.relative_position_bias(p[prefix + "relative_position_bias_table"], index).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    P = torch.softmax(s, dim=-1)
    o = (P @ v).transpose(1, 2).reshape(B_, N, C)                      # merged heads, input of proj

    g: Grads = {}
    do, g[prefix + "proj.weight"], g[prefix + "proj.bias"] = linear_bwd(o, Wp, dout)
    do = do.reshape(B_, N, heads, hd).permute(0, 2, 1, 3)             # (B_, h, N, hd)
    dv = P.transpose(-2, -1) @ do                                      # dV = P^T dO
    dP = do @ v.transpose(-2, -1)                                      # dP = dO V^T
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))                     # softmax backward, row-wise
    dq = (dS @ k) * scale                                              # q was scaled before QK^T (model.py:497)
    dk = dS.transpose(-2, -1) @ q
    # relative-position bias: every (window, batch) adds the same table entry -> scatter-add over the index
    dbias = dS.sum(0)                                                  # (heads, N, N)
    dtab = torch.zeros_like(p[prefix + "relative_position_bias_table"])
    dtab.index_add_(0, index.reshape(-1), dbias.permute(1, 2, 0).reshape(N * N, heads))
    g[prefix + "relative_position_bias_table"] = dtab
    merge = lambda t: t.permute(0, 2, 1, 3).reshape(B_, N, C)          # noqa: E731
    dxq, g[prefix + "qkv.to_q.weight"], g[prefix + "qkv.to_q.bias"] = linear_bwd(x, Wq, merge(dq))
    dxkv, g[prefix + "qkv.to_kv.weight"], g[prefix + "qkv.to_kv.bias"] = linear_bwd(x, Wkv, torch.cat([merge(dk), merge(dv)], -1))
    return dxq + dxkv, g


# ----------------------------------------------------------------------------------------
# LeFF (model.py:666-685)
# ----------------------------------------------------------------------------------------
def dwconv3x3_bwd(h: Tensor, w: Tensor, dc: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Depthwise 3x3, zero padding 1, on channel-last maps.  h, dc: (B, H, W, ch); w: (ch, 1, 3, 3).
    c[y,x] = b + sum_{ky,kx} w[ky,kx] h[y+ky-1, x+kx-1]
      dh[y,x] = sum_{ky,kx} w[ky,kx] dc[y-ky+1, x-kx+1]            (correlation with the flipped taps)
      dw[ky,kx] = sum_{b,y,x} dc[y,x] h[y+ky-1, x+kx-1];  db = sum dc."""
    B, H, W, ch = h.shape
    hp = F.pad(h, (0, 0, 1, 1, 1, 1))
    dcp = F.pad(dc, (0, 0, 1, 1, 1, 1))
    dh = torch.zeros_like(h)
    dw = torch.zeros_like(w)
    for ky in range(3):
        for kx in range(3):
            dh += w[:, 0, ky, kx] * dcp[:, 2 - ky:2 - ky + H, 2 - kx:2 - kx + W, :]
            dw[:, 0, ky, kx] = (dc * hp[:, ky:ky + H, kx:kx + W, :]).sum((0, 1, 2))
    return dh, dw, dc.sum((0, 1, 2))


def leff_bwd(z: Tensor, p: Dict[str, Tensor], prefix: str, dout: Tensor) -> Tuple[Tensor, Grads]:
    """z: (B, L, C) = LN2 output; dout: gradient of the linear2 output.  Recomputes a1, h1, c, g2 from z."""
    B, L, C = z.shape
    hh = int(math.sqrt(L))
    W1, W2, Wd = p[prefix + "linear1.0.weight"], p[prefix + "linear2.0.weight"], p[prefix + "dwconv.0.weight"]
    a1 = z @ W1.t() + p[prefix + "linear1.0.bias"]
    h1 = O.gelu_erf(a1).reshape(B, hh, hh, -1)                          # the tensor leff2 reads from HBM in the forward
    c = F.conv2d(h1.permute(0, 3, 1, 2), Wd, p[prefix + "dwconv.0.bias"], padding=1, groups=h1.shape[-1]).permute(0, 2, 3, 1)
    g2 = O.gelu_erf(c).reshape(B, L, -1)
    g: Grads = {}
    dg2, g[prefix + "linear2.0.weight"], g[prefix + "linear2.0.bias"] = linear_bwd(g2, W2, dout)
    dc = dg2.reshape(B, hh, hh, -1) * gelu_erf_grad(c)
    dh1, g[prefix + "dwconv.0.weight"], g[prefix + "dwconv.0.bias"] = dwconv3x3_bwd(h1, Wd, dc)
    da1 = dh1.reshape(B, L, -1) * gelu_erf_grad(a1)
    dz, g[prefix + "linear1.0.weight"], g[prefix + "linear1.0.bias"] = linear_bwd(z, W1, da1)
    return dz, g


# ----------------------------------------------------------------------------------------
# LeWin block (model.py:908-989)
# ----------------------------------------------------------------------------------------
def lewin_block_bwd(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dy: Tensor, win: int = WIN,
                    mask: Optional[Tensor] = None) -> Tuple[Tensor, Grads]:
    """Backward of one block given only its input x (B, L, C) and dy = dL/d(output).  Eval mode (DropPath = identity)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    attn_mask = O.input_attn_mask(mask, H, W, win) if mask is not None else None
    if shift > 0:
        sm = O.shift_attn_mask(H, W, win, shift)
        attn_mask = attn_mask + sm if attn_mask is not None else sm
    n1w, n2w = p[prefix + "norm1.weight"], p[prefix + "norm2.weight"]
    has_mod = (prefix + "modulator.weight") in p

    def to_windows(t: Tensor) -> Tensor:        # roll(-shift) + window_partition: a pure permutation (model.py:957,962)
        t = t.reshape(B, H, W, C)
        if shift > 0:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        return O.window_partition(t, win).reshape(-1, win * win, C)

    def from_windows(t: Tensor) -> Tensor:      # its inverse: window_reverse + roll(+shift) (model.py:975-980)
        t = O.window_reverse(t.reshape(-1, win, win, C), win, H, W)
        if shift > 0:
            t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
        return t.reshape(B, L, C)

    # ---- recompute the forward of the attention half (kernel 1) up to x1
    yw = to_windows(O.layer_norm(x, n1w, p[prefix + "norm1.bias"]))
    if has_mod:
        yw = yw + p[prefix + "modulator.weight"]
    x1 = x + from_windows(O.window_attention(yw, p, prefix + "attn.", heads, attn_mask))
    z = O.layer_norm(x1, n2w, p[prefix + "norm2.bias"])

    g: Grads = {}
    # ---- second residual branch: x2 = x1 + leff(LN2(x1))                                   (model.py:987)
    dz, gm = leff_bwd(z, p, prefix + "mlp.", dy)
    g.update(gm)
    dx1, g[prefix + "norm2.weight"], g[prefix + "norm2.bias"] = layer_norm_bwd(x1, n2w, dz)
    dx1 = dx1 + dy
    # ---- first residual branch: x1 = x + from_windows(attn(to_windows(LN1(x)) + modulator))    (model.py:951-986)
    dyw, ga = window_attention_bwd(yw, p, prefix + "attn.", heads, attn_mask, to_windows(dx1))   # grad wrt window rows
    g.update(ga)
    if has_mod:                                   # the (64, C) table is added to EVERY window: sum over windows
        g[prefix + "modulator.weight"] = dyw.sum(0)
    dx, g[prefix + "norm1.weight"], g[prefix + "norm1.bias"] = layer_norm_bwd(x, n1w, from_windows(dyw))
    return dx + dx1, g


# ----------------------------------------------------------------------------------------
# samplers, stem, head
# ----------------------------------------------------------------------------------------
def _tok2img(x: Tensor) -> Tensor:
    B, L, C = x.shape
    H = int(math.sqrt(L))
    return x.transpose(1, 2).reshape(B, C, H, H)


def _img2tok(y: Tensor) -> Tensor:
    return y.flatten(2).transpose(1, 2).contiguous()


def conv2d_bwd(x: Tensor, w: Tensor, dy: Tensor, stride: int, padding: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Dense Conv2d backward as GEMMs on the unfolded input (the shape the HIP implicit-GEMM kernels use):
    cols (B, Cin*kh*kw, P);  dW = dy_flat cols^T;  dx = fold(W^T dy_flat);  db = sum dy."""
    Cout, Cin, kh, kw = w.shape
    cols = F.unfold(x, (kh, kw), padding=padding, stride=stride)                    # (B, Cin*kh*kw, P)
    dyf = dy.flatten(2)                                                             # (B, Cout, P)
    dw = torch.einsum("bop,bkp->ok", dyf, cols).reshape(w.shape)
    dcols = torch.einsum("ok,bop->bkp", w.reshape(Cout, -1), dyf)
    dx = F.fold(dcols, x.shape[-2:], (kh, kw), padding=padding, stride=stride)
    return dx, dw, dy.sum((0, 2, 3))


def downsample_bwd(x: Tensor, p: Dict[str, Tensor], prefix: str, dy: Tensor) -> Tuple[Tensor, Grads]:
    """Conv2d(C,2C,k4,s2,p1) on tokens (model.py:739-746)."""
    dxi, dw, db = conv2d_bwd(_tok2img(x), p[prefix + "conv.0.weight"], _tok2img(dy), 2, 1)
    return _img2tok(dxi), {prefix + "conv.0.weight": dw, prefix + "conv.0.bias": db}


def upsample_bwd(x: Tensor, p: Dict[str, Tensor], prefix: str, dy: Tensor) -> Tuple[Tensor, Grads]:
    """ConvTranspose2d(Cin,Cout,k2,s2) on tokens (model.py:765-771): non-overlapping, i.e. 4 independent 1x1 GEMMs.
    y[b,co,2i+dy,2j+dx] = bias[co] + sum_ci x[b,ci,i,j] w[ci,co,dy,dx]."""
    w = p[prefix + "deconv.0.weight"]                                               # (Cin, Cout, 2, 2)
    xi, dyi = _tok2img(x), _tok2img(dy)
    B, Cin, H, W = xi.shape
    d4 = dyi.reshape(B, -1, H, 2, W, 2)                                             # (B, Cout, i, dy, j, dx)
    dx = torch.einsum("boiyjx,coyx->bcij", d4, w)
    dw = torch.einsum("bcij,boiyjx->coyx", xi, d4)
    return _img2tok(dx), {prefix + "deconv.0.weight": dw, prefix + "deconv.0.bias": dyi.sum((0, 2, 3))}


def input_proj_bwd(x: Tensor, p: Dict[str, Tensor], dy: Tensor) -> Tuple[Tensor, Grads]:
    """conv3x3 + LeakyReLU(0.01) -> tokens (model.py:795-800)."""
    w = p["input_proj.proj.0.weight"]
    pre = F.conv2d(x, w, p["input_proj.proj.0.bias"], stride=1, padding=1)
    dpre = _tok2img(dy) * torch.where(pre >= 0, torch.ones_like(pre), torch.full_like(pre, 0.01))
    dx, dw, db = conv2d_bwd(x, w, dpre, 1, 1)
    return dx, {"input_proj.proj.0.weight": dw, "input_proj.proj.0.bias": db}


def output_proj_bwd(x: Tensor, p: Dict[str, Tensor], dy: Tensor) -> Tuple[Tensor, Grads]:
    """tokens -> conv3x3 (model.py:828-836); dy is an image gradient (B,3,H,W)."""
    dxi, dw, db = conv2d_bwd(_tok2img(x), p["output_proj.proj.0.weight"], dy, 1, 1)
    return _img2tok(dxi), {"output_proj.proj.0.weight": dw, "output_proj.proj.0.bias": db}


# ----------------------------------------------------------------------------------------
# whole model (model.py:1269-1305): forward keeping every BLOCK INPUT and sampler input, then the reverse sweep
# ----------------------------------------------------------------------------------------
def charbonnier_loss_bwd(y: Tensor, target: Tensor, eps: float = 1e-3) -> Tensor:
    """d mean(sqrt(d^2+eps^2)) / dy = d / sqrt(d^2+eps^2) / numel   (losses.py:41-52)."""
    d = y - target
    return d / torch.sqrt(d * d + eps * eps) / d.numel()


def uformer_backward(x: Tensor, p: Dict[str, Tensor], dy: Tensor, *, img_size: int, embed_dim: int, depths: Sequence[int],
                     num_heads: Sequence[int], win: int = WIN, dd_in: int = 3) -> Tuple[Tensor, Grads]:
    """Gradient of the network output wrt the input image and every parameter, given dy = dL/d(output image)."""
    shifts = O.block_shifts(img_size, depths, win)
    saved: Dict[str, Tensor] = {}

    def stage_fwd(s: int, t: Tensor) -> Tensor:
        for i in range(depths[s]):
            saved[f"{O.STAGES[s]}.{i}"] = t
            t = O.lewin_block(t, p, f"{O.STAGES[s]}.blocks.{i}.", num_heads[s], shifts[s][i], win)
        return t

    def stage_bwd(s: int, d: Tensor, g: Grads) -> Tensor:
        for i in reversed(range(depths[s])):
            d, gb = lewin_block_bwd(saved[f"{O.STAGES[s]}.{i}"], p, f"{O.STAGES[s]}.blocks.{i}.", num_heads[s], shifts[s][i], d, win)
            g.update(gb)
        return d

    # ---- forward, keeping block inputs, sampler inputs and the skip widths
    t = O.input_proj(x, p)
    skips = []
    for s in range(4):
        t = stage_fwd(s, t)
        skips.append(t)
        saved[f"down{s}"] = t
        t = O.downsample(t, p, f"dowsample_{s}.")
    t = stage_fwd(4, t)
    for k in range(4):
        saved[f"up{k}"] = t
        t = torch.cat([O.upsample(t, p, f"upsample_{k}."), skips[3 - k]], -1)      # model.py:1288
        t = stage_fwd(5 + k, t)
    saved["head"] = t

    # ---- reverse sweep
    g: Grads = {}
    d, gh = output_proj_bwd(saved["head"], p, dy)                                    # y = x + conv(...) when dd_in == 3
    g.update(gh)
    dskip = [None] * 4
    for k in reversed(range(4)):
        d = stage_bwd(5 + k, d, g)
        cup = d.shape[-1] - skips[3 - k].shape[-1]
        dskip[3 - k] = d[..., cup:]                                                  # gradient of the concat's skip half
        d, gu = upsample_bwd(saved[f"up{k}"], p, f"upsample_{k}.", d[..., :cup].contiguous())
        g.update(gu)
    d = stage_bwd(4, d, g)
    for s in reversed(range(4)):
        d, gd = downsample_bwd(saved[f"down{s}"], p, f"dowsample_{s}.", d)
        g.update(gd)
        d = stage_bwd(s, d + dskip[s], g)
    dx, gi = input_proj_bwd(x, p, d)
    g.update(gi)
    if dd_in == 3:
        dx = dx + dy                                                                 # global residual, model.py:1305
    return dx, g
