"""TEST / CALIBRATION INFRASTRUCTURE -- never imported by the product (uformer_amd/).

The reference's op sequence (``/root/reference/model.py``) written with the library ops the reference's own ``nn.Module``s
dispatch to -- ``F.linear`` (nn.Linear: hipBLASLt / rocBLAS on ROCm), ``F.layer_norm`` (nn.LayerNorm), ``F.gelu`` (nn.GELU, erf form),
``softmax``, ``F.conv2d`` / ``F.conv_transpose2d`` (MIOpen) -- and device-agnostic, so that ``bench.py``'s baseline leg can time
"the reference forward on the vendor stack of the SAME box" (PyTorch-ROCm on the MI355X, fp32 and under bf16 autocast) next to the
hand-written HIP path.  It is a calibration figure (VERDICT r04 "next" 5), not the oracle: ``oracle/uformer_oracle.py`` stays the
parity checker; ``tests/test_oracle_golden.py::test_vendor_forward_equals_oracle`` pins this file to it on the CPU.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import uformer_oracle as O

Tensor = torch.Tensor
WIN = O.WIN


def _window_attention(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, bias: Tensor, mask: Optional[Tensor]) -> Tensor:
    """model.py:494-522 + LinearProjection :431-442.  ``bias``: (heads, N, N) gathered once per block (:500-502)."""
    B_, N, C = x.shape
    hd = C // heads
    q = F.linear(x, p[pre + "qkv.to_q.weight"], p[pre + "qkv.to_q.bias"]).reshape(B_, N, 1, heads, hd).permute(2, 0, 3, 1, 4)[0]
    kv = F.linear(x, p[pre + "qkv.to_kv.weight"], p[pre + "qkv.to_kv.bias"]).reshape(B_, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]                                              # :439-442
    attn = (q * (hd ** -0.5)) @ k.transpose(-2, -1)                   # :497-498
    attn = attn + bias.unsqueeze(0).to(attn.dtype)                    # :506
    if mask is not None:                                              # :508-512
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0).to(attn.dtype)).view(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)                                # :513-515
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)                # :519
    return F.linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])  # :520


def _leff(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """model.py:666-685 (modules :657-661)."""
    B, L, C = x.shape
    hh = int(math.sqrt(L))
    h = F.gelu(F.linear(x, p[pre + "linear1.0.weight"], p[pre + "linear1.0.bias"]))
    hid = h.shape[-1]
    h = h.reshape(B, hh, hh, hid).permute(0, 3, 1, 2)                 # :674 (the reference does not make it contiguous either)
    h = F.gelu(F.conv2d(h, p[pre + "dwconv.0.weight"], p[pre + "dwconv.0.bias"], stride=1, padding=1, groups=hid))
    h = h.permute(0, 2, 3, 1).reshape(B, L, hid)                      # :680
    return F.linear(h, p[pre + "linear2.0.weight"], p[pre + "linear2.0.bias"])


def _block(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, shift: int, masks: Dict[tuple, Tensor]) -> Tensor:
    """model.py:908-989, eval mode, no user mask."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    mask = None
    if shift > 0:                                                     # :924-942 (the reference rebuilds it every forward; cached here per (H, device))
        key = (H, W, shift, x.device)
        if key not in masks:
            masks[key] = O.shift_attn_mask(H, W, WIN, shift).to(x.device)
        mask = masks[key]
    y = F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5).view(B, H, W, C)   # :952
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))       # :957
    yw = O.window_partition(y, WIN).reshape(-1, WIN * WIN, C)         # :962-963
    if (pre + "modulator.weight") in p:                               # :966-969
        yw = yw + p[pre + "modulator.weight"]
    idx = p[pre + "attn.relative_position_index"].reshape(-1)
    bias = p[pre + "attn.relative_position_bias_table"][idx].reshape(WIN * WIN, WIN * WIN, -1).permute(2, 0, 1).contiguous()   # :500-502
    aw = _window_attention(yw, p, pre + "attn.", heads, bias, mask)   # :972
    y = O.window_reverse(aw.reshape(-1, WIN, WIN, C), WIN, H, W)      # :975-976
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))         # :980
    x = x + y.reshape(B, L, C)                                        # :986
    return x + _leff(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5), p, pre + "mlp.")   # :987


def forward(x: Tensor, p: Dict[str, Tensor], *, img_size: int, embed_dim: int, depths: Sequence[int], num_heads: Sequence[int],
            dd_in: int = 3) -> Tensor:
    """Uformer.forward, model.py:1269-1305, on whatever device ``x`` and ``p`` live on."""
    shifts = O.block_shifts(img_size, depths, WIN)
    masks: Dict[tuple, Tensor] = {}

    def stage(y, s):
        for i in range(depths[s]):
            y = _block(y, p, f"{O.STAGES[s]}.blocks.{i}.", num_heads[s], shifts[s][i], masks)
        return y

    def tok2img(t):
        B, L, C = t.shape
        hh = int(math.sqrt(L))
        return t.transpose(1, 2).contiguous().view(B, C, hh, hh)      # :744, :769

    def img2tok(t):
        return t.flatten(2).transpose(1, 2).contiguous()              # :745, :770

    def down(t, k):
        return img2tok(F.conv2d(tok2img(t), p[f"dowsample_{k}.conv.0.weight"], p[f"dowsample_{k}.conv.0.bias"], stride=2, padding=1))

    def up(t, k):
        return img2tok(F.conv_transpose2d(tok2img(t), p[f"upsample_{k}.deconv.0.weight"], p[f"upsample_{k}.deconv.0.bias"], stride=2))

    y = img2tok(F.leaky_relu(F.conv2d(x, p["input_proj.proj.0.weight"], p["input_proj.proj.0.bias"], stride=1, padding=1), 0.01))   # :795-800
    c0 = stage(y, 0)
    c1 = stage(down(c0, 0), 1)
    c2 = stage(down(c1, 1), 2)
    c3 = stage(down(c2, 2), 3)
    c4 = stage(down(c3, 3), 4)
    d0 = stage(torch.cat([up(c4, 0), c3], -1), 5)                     # :1288
    d1 = stage(torch.cat([up(d0, 1), c2], -1), 6)
    d2 = stage(torch.cat([up(d1, 2), c1], -1), 7)
    d3 = stage(torch.cat([up(d2, 3), c0], -1), 8)
    out = F.conv2d(tok2img(d3), p["output_proj.proj.0.weight"], p["output_proj.proj.0.bias"], stride=1, padding=1)             # :828-836
    return x + out if dd_in == 3 else out                             # :1305
