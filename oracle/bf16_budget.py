"""Error budget of the 2-byte operand modes (bf16, and f16 = the reference's own AMP type), by source.  TEST / ANALYSIS INFRASTRUCTURE ONLY (same rules as uformer_oracle.py).

The HIP bf16 path rounds to bf16 at a fixed set of points and uses two approximations; this file restates the forward of
uformer_oracle.py with each of those as a SWITCH, so that ``bench.py --error-budget`` (and tests/test_host_logic.py) can turn
them on one at a time and attribute the max-abs / mean error of the restored image to its sources (VERDICT r01 "weak" 1).

Rounding points of the kernels, by switch name:
  w      every GEMM weight (q/kv/proj/linear1/linear2, Downsample, Upsample) is a bf16 operand
  xn     LN1(x) + modulator, the operand of the q/k/v projections            (attn_block phase 0 -> LDS)
  qkv    q * scale, k, v: MFMA accumulators repacked as bf16 operand fragments (attn_block phase 1)
  p      exp2(s - max): the UNNORMALISED probabilities are the bf16 operand of P.V; 1/sum is applied to O in f32
  o      the normalised head outputs, operand of proj                         (attn_block -> LDS)
  z      LN2(x), operand of linear1
  h1     GELU(linear1(z)), stored as bf16 in HBM (the tensor the depthwise conv reads)
  g2     GELU(dwconv(h1)), operand of linear2                                  (leff2 -> LDS)
  samp   the f32 stream rows read by Downsample / Upsample are converted to bf16 operands
  gelu   both GELUs in the sigmoid form x / (1 + 2^(x (A + B x^2))) instead of erf (uf_common.h gelu_bf2)
Everything else (residual stream, LayerNorm statistics, softmax max / sum, biases, stem, head, accumulation) is f32 in both.
"""
from __future__ import annotations

import math
from typing import Dict, FrozenSet, Iterable, Optional, Sequence

import torch
import torch.nn.functional as F

from . import uformer_oracle as O

Tensor = torch.Tensor
ALL = ("w", "xn", "qkv", "p", "o", "z", "h1", "g2", "samp", "gelu")


def bf(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def hf(x: Tensor) -> Tensor:
    """IEEE half rounding (round to nearest even, gradual underflow, saturation to inf above 65504: as v_cvt_f16_f32)."""
    return x.to(torch.float16).to(torch.float32)


ROUND = {"bf16": bf, "f16": hf}


def gelu_sigmoid(x: Tensor) -> Tensor:
    a = -2.3022081985
    b = -0.10294324
    return x / (1.0 + torch.exp2(x * (a + b * x * x)))


class Budget:
    def __init__(self, on: Iterable[str], operand: str = "bf16"):
        self.rnd = ROUND[operand]
        self.on: FrozenSet[str] = frozenset(on)
        bad = self.on - set(ALL)
        if bad:
            raise ValueError(f"unknown switches {sorted(bad)}")

    def r(self, name: str, x: Tensor) -> Tensor:
        return self.rnd(x) if name in self.on else x

    def gelu(self, x: Tensor) -> Tensor:
        return gelu_sigmoid(x) if "gelu" in self.on else O.gelu_erf(x)


def _attn(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, mask: Optional[Tensor], q: Budget) -> Tensor:
    B_, N, C = x.shape
    hd = C // heads
    W = lambda k: q.r("w", p[prefix + k])                                        # noqa: E731
    qq = x @ W("qkv.to_q.weight").t() + p[prefix + "qkv.to_q.bias"]
    kv = x @ W("qkv.to_kv.weight").t() + p[prefix + "qkv.to_kv.bias"]
    qq = qq.reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    k = kv[..., :C].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    v = kv[..., C:].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)
    # the kernel folds log2(e) into the q scale and the bias table and takes exp2: same value, the rounding of q happens in
    # that scaled domain (a power-of-two-free constant, so it is a different -- equally sized -- rounding; emulated as such)
    l2e = math.log2(math.e)
    qq = q.r("qkv", qq * (hd ** -0.5) * l2e)
    k, v = q.r("qkv", k), q.r("qkv", v)
    s = qq @ k.transpose(-2, -1)
    bias = O.relative_position_bias(p[prefix + "relative_position_bias_table"], p[prefix + "relative_position_index"]) * l2e
    s = s + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, heads, N, N) + (mask * l2e).unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    e = torch.exp2(s - s.amax(-1, keepdim=True))
    o = (q.r("p", e) @ v) / e.sum(-1, keepdim=True)
    o = q.r("o", o.transpose(1, 2).reshape(B_, N, C))
    return o @ W("proj.weight").t() + p[prefix + "proj.bias"]


def _leff(z: Tensor, p: Dict[str, Tensor], prefix: str, q: Budget) -> Tensor:
    B, L, C = z.shape
    hh = int(math.sqrt(L))
    h = q.r("h1", q.gelu(z @ q.r("w", p[prefix + "linear1.0.weight"]).t() + p[prefix + "linear1.0.bias"]))
    hid = h.shape[-1]
    h = F.conv2d(h.reshape(B, hh, hh, hid).permute(0, 3, 1, 2), p[prefix + "dwconv.0.weight"], p[prefix + "dwconv.0.bias"], stride=1, padding=1, groups=hid)
    h = q.r("g2", q.gelu(h)).permute(0, 2, 3, 1).reshape(B, L, hid)
    return h @ q.r("w", p[prefix + "linear2.0.weight"]).t() + p[prefix + "linear2.0.bias"]


def _block(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int, q: Budget) -> Tensor:
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    mask = O.shift_attn_mask(H, W, O.WIN, shift) if shift > 0 else None
    y = O.layer_norm(x, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"]).reshape(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = O.window_partition(y, O.WIN).reshape(-1, 64, C)
    if (prefix + "modulator.weight") in p:
        yw = yw + p[prefix + "modulator.weight"]
    aw = _attn(q.r("xn", yw), p, prefix + "attn.", heads, mask, q)
    y = O.window_reverse(aw.reshape(-1, O.WIN, O.WIN, C), O.WIN, H, W)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = x + y.reshape(B, L, C)
    z = q.r("z", O.layer_norm(x, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]))
    return x + _leff(z, p, prefix + "mlp.", q)


def _down(x: Tensor, p: Dict[str, Tensor], prefix: str, q: Budget) -> Tensor:
    B, L, C = x.shape
    H = int(math.sqrt(L))
    y = F.conv2d(q.r("samp", x).transpose(1, 2).reshape(B, C, H, H), q.r("w", p[prefix + "conv.0.weight"]), p[prefix + "conv.0.bias"], stride=2, padding=1)
    return y.flatten(2).transpose(1, 2).contiguous()


def _up(x: Tensor, p: Dict[str, Tensor], prefix: str, q: Budget) -> Tensor:
    B, L, C = x.shape
    H = int(math.sqrt(L))
    y = F.conv_transpose2d(q.r("samp", x).transpose(1, 2).reshape(B, C, H, H), q.r("w", p[prefix + "deconv.0.weight"]), p[prefix + "deconv.0.bias"], stride=2)
    return y.flatten(2).transpose(1, 2).contiguous()


@torch.no_grad()
def forward(x: Tensor, p: Dict[str, Tensor], switches: Iterable[str], *, img_size: int, embed_dim: int, depths: Sequence[int],
            num_heads: Sequence[int], dd_in: int = 3, operand: str = "bf16") -> Tensor:
    """uformer_oracle.uformer_forward with the roundings / approximations named in ``switches`` applied (eval mode, no mask);
    ``operand`` = the 2-byte type the rounding points use ("bf16" or "f16")."""
    q = Budget(switches, operand)
    shifts = O.block_shifts(img_size, depths)

    def stage(y: Tensor, s: int) -> Tensor:
        for i in range(depths[s]):
            y = _block(y, p, f"{O.STAGES[s]}.blocks.{i}.", num_heads[s], shifts[s][i], q)
        return y

    y = O.input_proj(x, p)
    skips = []
    for s in range(4):
        y = stage(y, s)
        skips.append(y)
        y = _down(y, p, f"dowsample_{s}.", q)
    y = stage(y, 4)
    for k in range(4):
        y = stage(torch.cat([_up(y, p, f"upsample_{k}.", q), skips[3 - k]], -1), 5 + k)
    y = O.output_proj(y, p)
    return x + y if dd_in == 3 else y


def error_budget(x: Tensor, p: Dict[str, Tensor], exact: Tensor, groups=None, **kw) -> Dict[str, Dict[str, float]]:
    """One switch (or group of switches) at a time, then all together: max-abs and mean signed / absolute error of the restored
    image against ``exact`` (the all-f32 oracle output)."""
    groups = groups or [(s,) for s in ALL] + [("xn", "qkv", "p", "o", "z", "h1", "g2", "samp"), ALL]
    out = {}
    for gsw in groups:
        y = forward(x, p, gsw, **kw)
        d = y - exact
        name = "+".join(gsw) if len(gsw) < len(ALL) else "all"
        out[name] = {"max_abs": float(d.abs().max()), "mean_abs": float(d.abs().mean()), "mean_signed": float(d.mean())}
    return out
