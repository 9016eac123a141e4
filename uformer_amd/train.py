"""Backward of the whole path assembled from the C-ABI building blocks (first end-to-end form).

Every FLOP-carrying step is a HIP kernel behind the C ABI (LayerNorm fwd/bwd, the projections and their input / weight
gradients, window attention fwd/bwd, the depthwise stencil in both directions and its tap gradients, GELU'); PyTorch
only permutes layouts (head merge, window order), adds residuals and scatter-adds the 64x64 bias gradient into the
225-row table, and unfolds / folds the 3x3 and 4x4 convolution patches of the stem, head and Downsample (4 % of
the FLOPs) so that they, too, go through the GEMM kernels -- plumbing.  This is the op-by-op form: it keeps the intermediates of the forward instead of recomputing
them inside fused backward kernels (DESIGN.md section 7 has the fused plan); it exists to make the block's gradients
exact and testable against the reference's autograd before the fused kernels are written.

Eval-mode semantics (DropPath = identity), as the gradient fixtures.  model.py:908-989.
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
# UF_TRAIN_SEPARATE_GELU=1: use uf_gelu_fwd instead of running linear1 and the depthwise conv twice in the training forward
# (validated on the GPU by tests/test_gpu_bwd.py; off by default until the training step has been re-timed with it)
_SEPARATE_GELU = os.environ.get("UF_TRAIN_SEPARATE_GELU", "0") == "1"
Grads = Dict[str, Tensor]


def _zeros(n: int, dev) -> Tensor:
    return torch.zeros(n, device=dev)


def _input_grad(dy: Tensor, w: Tensor) -> Tensor:
    """dX = dY W for y = x W^T + b: the forward GEMM with the transposed weight and a zero bias."""
    return ops.linear(dy, w.t().contiguous(), _zeros(w.shape[1], dy.device))


# ------------------------------------------------------------------------------------------------------------------
# LeWin block (model.py:908-989)
# ------------------------------------------------------------------------------------------------------------------
def lewin_block_forward(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dtype: torch.dtype,
                        drop: Optional[Tensor] = None) -> Tuple[Tensor, Saved]:
    """x: (B, L, C) f32 on the GPU -> (y, saved).  Op-by-op forward that keeps what the backward reads.
    ``drop``: None (eval) or (2, B) per-sample DropPath scales bernoulli(keep)/keep of the two residual branches (model.py:986-987)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    M, hd = B * L, C // heads
    T = dtype
    f = lambda k: p[prefix + k]                                             # noqa: E731
    x2 = x.reshape(M, C).float().contiguous()
    mod = f("modulator.weight") if (prefix + "modulator.weight") in p else None
    wqkv = torch.cat([f("attn.qkv.to_q.weight"), f("attn.qkv.to_kv.weight")], 0).to(T)
    bqkv = torch.cat([f("attn.qkv.to_q.bias"), f("attn.qkv.to_kv.bias")], 0)
    wp, w1, w2 = f("attn.proj.weight").to(T), f("mlp.linear1.0.weight").to(T), f("mlp.linear2.0.weight").to(T)
    w9 = packing.pack_dwconv(f("mlp.dwconv.0.weight"))
    bias = packing.rpb_dense(f("attn.relative_position_bias_table"), f("attn.relative_position_index"))
    xn = ops.layernorm(x2, f("norm1.weight"), f("norm1.bias"), B=B, H=H, W=W, dtype=T, windowed=True, shift=shift, modulator=mod)
    q, k, vt = ops.qkv(xn, wqkv, bqkv, heads)                                # window rows; q already scaled
    o = ops.window_attention_core(q, k, vt, bias, H=H, W=W, shift=shift)     # (M, C) window rows
    yw = ops.linear(o, wp, f("attn.proj.bias"))
    s1 = drop[0].float().repeat_interleave(L).reshape(M, 1) if drop is not None else None      # per-token copy of the per-sample scale
    s2 = drop[1].float().repeat_interleave(L).reshape(M, 1) if drop is not None else None
    br1 = ops.window_reverse(yw.reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C).float()
    x1 = x2 + (br1 * s1 if s1 is not None else br1)
    z = ops.layernorm(x1, f("norm2.weight"), f("norm2.bias"), B=B, H=H, W=W, dtype=T)
    a1 = ops.linear(z, w1, f("mlp.linear1.0.bias"))                          # pre-activation, kept for GELU'
    c_bias = f("mlp.dwconv.0.bias")
    if _SEPARATE_GELU:                                                      # one GEMM / one stencil + an elementwise GELU pass each
        h1 = ops.gelu(a1).reshape(B, H, W, 4 * C)
        c = ops.dwconv3x3(h1, w9, c_bias, gelu=False)                       # pre-activation of the second GELU
        g2 = ops.gelu(c).reshape(M, 4 * C)
    else:                                                                   # activated copies recomputed by the fused-epilogue kernels
        h1 = ops.linear(z, w1, f("mlp.linear1.0.bias"), act=1).reshape(B, H, W, 4 * C)
        c = ops.dwconv3x3(h1, w9, c_bias, gelu=False)
        g2 = ops.dwconv3x3(h1, w9, c_bias, gelu=True).reshape(M, 4 * C)
    br2 = ops.linear(g2, w2, f("mlp.linear2.0.bias")).float()
    y = x1 + (br2 * s2 if s2 is not None else br2)
    saved = dict(s1=s1, s2=s2, p=p, prefix=prefix, heads=heads, shift=shift, T=T, shape=(B, L, C), x2=x2, xn=xn, q=q, k=k, vt=vt, o=o, x1=x1, z=z, a1=a1,
                 h1=h1, c=c, g2=g2, wqkv=wqkv, wp=wp, w1=w1, w2=w2, w9=w9, bias=bias, mod=mod is not None)
    return y.reshape(B, L, C), saved


def lewin_block_backward(sv: Saved, dy: Tensor) -> Tuple[Tensor, Grads]:
    """dy: (B, L, C) gradient of the block output -> (dx, gradients keyed like the reference's named_parameters())."""
    p, prefix, heads, shift, T = sv["p"], sv["prefix"], sv["heads"], sv["shift"], sv["T"]
    B, L, C = sv["shape"]
    H = W = int(math.sqrt(L))
    M, hd = B * L, C // heads
    f = lambda k: p[prefix + k]                                             # noqa: E731
    g: Grads = {}
    dyf = dy.reshape(M, C).float()
    dyT = (dyf * sv["s2"] if sv["s2"] is not None else dyf).to(T).contiguous()      # gradient entering the (scaled) LeFF branch
    # LeFF: linear2 -> GELU -> depthwise -> GELU -> linear1                                   (model.py:666-685)
    g[prefix + "mlp.linear2.0.weight"], g[prefix + "mlp.linear2.0.bias"] = ops.linear_wgrad(dyT, sv["g2"])
    dg2 = _input_grad(dyT, sv["w2"])
    dc = ops.gelu_bwd(sv["c"].reshape(M, 4 * C), dg2).reshape(B, H, W, 4 * C)
    dw9, g[prefix + "mlp.dwconv.0.bias"] = ops.dwconv3x3_wgrad(sv["h1"], dc)
    g[prefix + "mlp.dwconv.0.weight"] = dw9.t().reshape(4 * C, 1, 3, 3)
    dh1 = ops.dwconv3x3(dc, sv["w9"].flip(0).contiguous(), None, gelu=False)   # input gradient = flipped-tap stencil
    da1 = ops.gelu_bwd(sv["a1"], dh1.reshape(M, 4 * C))
    g[prefix + "mlp.linear1.0.weight"], g[prefix + "mlp.linear1.0.bias"] = ops.linear_wgrad(da1, sv["z"])
    dz = _input_grad(da1, sv["w1"]).float()
    dx1, g[prefix + "norm2.weight"], g[prefix + "norm2.bias"] = ops.layernorm_bwd(sv["x1"], f("norm2.weight"), dz)
    dx1 = dx1 + dy.reshape(M, C).float()
    # attention half: proj -> attention -> qkv -> (+modulator) -> partition/roll -> LN1              (model.py:951-986)
    dbr1 = dx1 * sv["s1"] if sv["s1"] is not None else dx1                          # gradient entering the (scaled) attention branch
    dyw = ops.window_partition(dbr1.reshape(B, H, W, C), 8, shift).reshape(M, C).to(T)
    g[prefix + "attn.proj.weight"], g[prefix + "attn.proj.bias"] = ops.linear_wgrad(dyw, sv["o"])
    do = _input_grad(dyw, sv["wp"])
    dq, dk, dvt, dbias = ops.window_attention_bwd(sv["q"], sv["k"], sv["vt"], sv["bias"], do, H, W, shift)
    index = f("attn.relative_position_index")
    dtab = torch.zeros_like(f("attn.relative_position_bias_table"), dtype=torch.float32)
    dtab.index_add_(0, index.reshape(-1), dbias.permute(1, 2, 0).reshape(64 * 64, heads))
    g[prefix + "attn.relative_position_bias_table"] = dtab
    nW = M // 64
    merge = lambda t: t.reshape(nW, heads, 64, hd).permute(0, 2, 1, 3).reshape(M, C)      # noqa: E731  (nW,h,64,hd) -> rows
    dqkv = torch.cat([merge(dq.float() * hd ** -0.5).to(T), merge(dk), merge(dvt.reshape(nW, heads, hd, 64).transpose(2, 3))], 1).contiguous()
    dWqkv, dbqkv = ops.linear_wgrad(dqkv, sv["xn"])
    g[prefix + "attn.qkv.to_q.weight"], g[prefix + "attn.qkv.to_kv.weight"] = dWqkv[:C], dWqkv[C:]
    g[prefix + "attn.qkv.to_q.bias"], g[prefix + "attn.qkv.to_kv.bias"] = dbqkv[:C], dbqkv[C:]
    dxn = _input_grad(dqkv, sv["wqkv"])
    if sv["mod"]:                                                           # the (64, C) table is added to every window
        g[prefix + "modulator.weight"] = dxn.float().reshape(nW, 64, C).sum(0)
    dln = ops.window_reverse(dxn.float().reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C)
    dx, g[prefix + "norm1.weight"], g[prefix + "norm1.bias"] = ops.layernorm_bwd(sv["x2"], f("norm1.weight"), dln)
    return (dx + dx1).reshape(B, L, C), g


def lewin_block_forward_backward(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, dy: Tensor,
                                 dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor, Grads]:
    y, sv = lewin_block_forward(x, p, prefix, heads, shift, dtype)
    dx, g = lewin_block_backward(sv, dy)
    return y, dx, g


# ------------------------------------------------------------------------------------------------------------------
# convolutions of the samplers / stem / head as patch GEMMs: dW = dY^T cols, dX = fold(dY W)   (4 % of the FLOPs)
# ------------------------------------------------------------------------------------------------------------------
def _pad_cols(t: Tensor, mult: int) -> Tensor:
    n = t.shape[-1]
    return t if n % mult == 0 else F.pad(t, (0, mult - n % mult))


def _conv_backward(x_img: Tensor, w: Tensor, dy_rows: Tensor, stride: int, padding: int, T: torch.dtype) -> Tuple[Tensor, Tensor, Tensor]:
    """x_img (B,Cin,H,W) f32, w (Cout,Cin,kh,kw), dy_rows (B*P, Cout) f32 in output-pixel order -> (dx_img, dW, db).
    The patch matrix is materialised by torch (unfold) and both products run through uf_linear_fwd / uf_linear_wgrad;
    column counts are padded to a multiple of 8 (zero columns) because the kernels move 16-byte pieces."""
    Cout, Cin, kh, kw = w.shape
    B = x_img.shape[0]
    cols = F.unfold(x_img, (kh, kw), padding=padding, stride=stride)               # (B, Cin*kh*kw, P)
    P, Kc = cols.shape[-1], cols.shape[1]
    colsT = _pad_cols(cols.transpose(1, 2).reshape(B * P, Kc), 8).to(T).contiguous()
    dyT = _pad_cols(dy_rows, 8).to(T).contiguous()
    wmat = _pad_cols(F.pad(w.reshape(Cout, Kc), (0, 0, 0, dyT.shape[1] - Cout)), 8).to(T)     # (Cout padded, Kc padded)
    dWm, db = ops.linear_wgrad(dyT, colsT)
    dcols = _input_grad(dyT, wmat).float()[:, :Kc]
    dx = F.fold(dcols.reshape(B, P, Kc).transpose(1, 2), x_img.shape[-2:], (kh, kw), padding=padding, stride=stride)
    return dx, dWm[:Cout, :Kc].reshape(w.shape), db[:Cout]


def _tok2img(x: Tensor, B: int) -> Tensor:
    L, C = x.shape[0] // B, x.shape[1]
    H = int(math.sqrt(L))
    return x.reshape(B, L, C).transpose(1, 2).reshape(B, C, H, H)


def _img2tok(y: Tensor) -> Tensor:
    B, C = y.shape[:2]
    return y.flatten(2).transpose(1, 2).reshape(-1, C).contiguous()


class UformerTape:
    """One forward of the whole model (model.py:1269-1305) that keeps what the reverse sweep reads, and that sweep.
    ``drop_scales``: None (eval semantics) or a (2 * n_blocks, B) tensor of DropPath scales in execution order."""

    def __init__(self, sd: Dict[str, Tensor], cfg, dtype: torch.dtype = torch.float32, drop_scales: Optional[Tensor] = None):
        self.sd, self.cfg, self.T, self.drop = sd, cfg, dtype, drop_scales

    def forward(self, img: Tensor) -> Tensor:
        from .spec import STAGES
        sd, cfg, T = self.sd, self.cfg, self.T
        B, _, H, W = img.shape
        self.B, self.H = B, H
        shifts = cfg.block_shifts()
        res = self.res = [H, H // 2, H // 4, H // 8, H // 16, H // 8, H // 4, H // 2, H]
        first = [sum(cfg.depths[:s]) for s in range(9)]
        self.saved_blocks: List[List[Saved]] = [[] for _ in range(9)]

        def stage_fwd(s: int, t: Tensor) -> Tensor:                             # t: (M, C) f32 token rows
            C = t.shape[1]
            t = t.reshape(B, res[s] * res[s], C)
            for i in range(cfg.depths[s]):
                bi = first[s] + i
                dr = self.drop[2 * bi:2 * bi + 2] if self.drop is not None else None
                t, sv = lewin_block_forward(t, sd, f"{STAGES[s]}.blocks.{i}.", cfg.num_heads[s], shifts[s][i], T, dr)
                self.saved_blocks[s].append(sv)
            return t.reshape(-1, C)

        # the samplers / stem / head run their inference kernels; their inputs are kept
        self.img = img
        t = ops.input_proj(img, packing.pack_input_proj(sd["input_proj.proj.0.weight"]), sd["input_proj.proj.0.bias"])
        self.stem_out = t
        self.skips, self.down_in, self.up_in = [], [], []
        for s in range(4):
            t = stage_fwd(s, t)
            self.skips.append(t)
            self.down_in.append(t)
            t = ops.downsample(t, packing.pack_downsample(sd[f"dowsample_{s}.conv.0.weight"], T), sd[f"dowsample_{s}.conv.0.bias"], B, res[s], res[s])
        t = stage_fwd(4, t)
        for k in range(4):
            self.up_in.append(t)
            up = ops.upsample(t, packing.pack_upsample(sd[f"upsample_{k}.deconv.0.weight"], T), sd[f"upsample_{k}.deconv.0.bias"], B, res[4 + k], res[4 + k])
            t = stage_fwd(5 + k, torch.cat([up, self.skips[3 - k]], 1))           # model.py:1288
        self.head_in = t
        return ops.output_proj(t, packing.pack_output_proj(sd["output_proj.proj.0.weight"]), sd["output_proj.proj.0.bias"], B, H, W,
                               img if cfg.dd_in == 3 else None)

    def backward(self, dy: Tensor) -> Tuple[Tensor, Grads]:
        sd, cfg, T, B, H, res = self.sd, self.cfg, self.T, self.B, self.H, self.res

        def stage_bwd(s: int, d: Tensor, g: Grads) -> Tensor:
            C = d.shape[1]
            d = d.reshape(B, res[s] * res[s], C)
            for sv in reversed(self.saved_blocks[s]):
                d, gb = lewin_block_backward(sv, d)
                g.update(gb)
            return d.reshape(-1, C)

        g: Grads = {}
        dy = dy.float()
        dy_rows = dy.permute(0, 2, 3, 1).reshape(B * H * H, 3)
        dxi, g["output_proj.proj.0.weight"], g["output_proj.proj.0.bias"] = _conv_backward(_tok2img(self.head_in, B), sd["output_proj.proj.0.weight"], dy_rows, 1, 1, T)
        d = _img2tok(dxi)
        dskip: List[Tensor] = [None] * 4
        for k in reversed(range(4)):
            d = stage_bwd(5 + k, d, g)
            Cs = self.skips[3 - k].shape[1]
            cup = d.shape[1] - Cs
            dskip[3 - k] = d[:, cup:]
            # ConvTranspose2d k2 s2 = four independent 1x1 GEMMs: gather the 2x2 output pixels of every input pixel into one row
            r = res[4 + k]
            w = sd[f"upsample_{k}.deconv.0.weight"]                                # (Cin, Cout, 2, 2)
            d4 = d[:, :cup].reshape(B, r, 2, r, 2, cup).permute(0, 1, 3, 2, 4, 5).reshape(B * r * r, 4 * cup).to(T).contiguous()
            wpk = packing.pack_upsample(w, T)                                      # (4*Cout, Cin), n = (dy*2+dx)*Cout + co
            dWp, dbp = ops.linear_wgrad(d4, self.up_in[k].to(T))
            g[f"upsample_{k}.deconv.0.weight"] = dWp.reshape(2, 2, cup, w.shape[0]).permute(3, 2, 0, 1).contiguous()
            g[f"upsample_{k}.deconv.0.bias"] = dbp.reshape(4, cup).sum(0)
            d = _input_grad(d4, wpk).float()
        d = stage_bwd(4, d, g)
        for s in reversed(range(4)):
            dxi, g[f"dowsample_{s}.conv.0.weight"], g[f"dowsample_{s}.conv.0.bias"] = _conv_backward(
                _tok2img(self.down_in[s], B), sd[f"dowsample_{s}.conv.0.weight"], d, 2, 1, T)
            d = stage_bwd(s, _img2tok(dxi) + dskip[s], g)
        so = self.stem_out
        dpre = d * torch.where(so >= 0, torch.ones_like(so), torch.full_like(so, 0.01))     # LeakyReLU(0.01), model.py:786
        dimg, g["input_proj.proj.0.weight"], g["input_proj.proj.0.bias"] = _conv_backward(self.img.float(), sd["input_proj.proj.0.weight"], dpre, 1, 1, T)
        if cfg.dd_in == 3:
            dimg = dimg + dy                                                      # global residual, model.py:1305
        return dimg, g


def uformer_forward_backward(img: Tensor, sd: Dict[str, Tensor], dy: Tensor, *, cfg, dtype: torch.dtype = torch.float32,
                             drop_scales: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Grads]:
    """Whole-model forward + backward.  img, dy: (B,3,H,W) f32 on the GPU; sd: the reference state_dict on the GPU; cfg:
    uformer_amd.spec.UformerConfig.  Returns (y, d img, parameter gradients keyed like named_parameters())."""
    tape = UformerTape(sd, cfg, dtype, drop_scales)
    y = tape.forward(img)
    dimg, g = tape.backward(dy)
    return y, dimg, g


class UformerFunction(torch.autograd.Function):
    """torch.autograd entry: ``y = UformerFunction.apply(img, cfg, dtype, drop_scales, names, *params)``; backward() hands the
    parameter gradients of the tape to autograd in the order of ``names`` (buffers such as relative_position_index get None)."""

    @staticmethod
    def forward(ctx, img, cfg, dtype, drop_scales, names, *params):
        sd = {n: p.detach() for n, p in zip(names, params)}
        tape = UformerTape(sd, cfg, dtype, drop_scales)
        y = tape.forward(img.detach().float().contiguous())
        ctx.tape, ctx.names = tape, names
        ctx.img_needs_grad = img.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        dimg, g = ctx.tape.backward(dy.contiguous())
        ctx.tape = None                                                           # free the saved activations
        grads = tuple(g.get(n) for n in ctx.names)
        return (dimg if ctx.img_needs_grad else None, None, None, None, None) + grads


def sample_drop_scales(rates: Sequence[float], B: int, device, generator: Optional[torch.Generator] = None) -> Tensor:
    """timm DropPath for every block in execution order: two rows (attention branch, LeFF branch) of per-sample scales
    bernoulli(1 - rate) / (1 - rate); rate 0 -> ones.  (model.py:883, :986-987; schedule :1093-1095.)"""
    rows = []
    for r in rates:
        for _ in range(2):
            if r <= 0.0:
                rows.append(torch.ones(B, device=device))
            else:
                keep = 1.0 - r
                rows.append(torch.bernoulli(torch.full((B,), keep, device=device), generator=generator) / keep)
    return torch.stack(rows)
