"""Host-side description of the reference's ``Uformer`` boundary: constructor kwargs,
the ``state_dict`` key layout and the per-stage geometry.

Pure Python / torch-CPU, no HIP.  References:
  * constructor kwargs            /root/reference model.py:1070-1077
  * arch factory (T/S/B kwargs)   utils/model_utils.py:56-81
  * shift / window clamp          model.py:863-866, :1030
  * state_dict layout             SURVEY.md Appendix C (probed from model.py)
"""
from __future__ import annotations

import dataclasses
import hashlib
import math
from typing import Dict, Iterator, List, Sequence, Tuple

import torch

STAGES = ("encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
          "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3")


@dataclasses.dataclass(frozen=True)
class UformerConfig:
    img_size: int = 256
    in_chans: int = 3
    dd_in: int = 3
    embed_dim: int = 32
    depths: Tuple[int, ...] = (2, 2, 2, 2, 2, 2, 2, 2, 2)
    num_heads: Tuple[int, ...] = (1, 2, 4, 8, 16, 16, 8, 4, 2)
    win_size: int = 8
    mlp_ratio: float = 4.0
    modulator: bool = False
    shift_flag: bool = True

    # ---- geometry -------------------------------------------------------------------
    def stage_dims(self) -> List[int]:
        """Channel width of the LeWin blocks of each of the 9 stages (model.py:1104-1245)."""
        e = self.embed_dim
        return [e, 2 * e, 4 * e, 8 * e, 16 * e, 16 * e, 8 * e, 4 * e, 2 * e]

    def stage_res_div(self) -> List[int]:
        """Resolution divisor of each stage relative to the input image."""
        return [1, 2, 4, 8, 16, 8, 4, 2, 1]

    def stage_has_modulator(self, s: int) -> bool:
        """Only decoder stages receive ``modulator`` (model.py:1197,1213,1229,1245)."""
        return self.modulator and s >= 5

    def block_shifts(self) -> List[List[int]]:
        """Constructor-time shift of every block (model.py:1030 then the clamp :863-866)."""
        out = []
        for s, d in enumerate(self.depths):
            res = self.img_size // self.stage_res_div()[s]
            row = []
            for i in range(d):
                sh = 0 if (i % 2 == 0 or not self.shift_flag) else self.win_size // 2
                if res <= self.win_size:
                    sh = 0
                row.append(sh)
            out.append(row)
        return out

    def upsample_io(self) -> List[Tuple[int, int]]:
        """(Cin, Cout) of upsample_0..3 (model.py:1182,1198,1214,1230)."""
        e = self.embed_dim
        return [(16 * e, 8 * e), (16 * e, 4 * e), (8 * e, 2 * e), (4 * e, e)]


def arch_config(name: str, img_size: int = 256, dd_in: int = 3) -> UformerConfig:
    """kwargs of ``utils.get_arch`` (utils/model_utils.py:65-78); 'tiny' is BASELINE config 0."""
    if name == "Uformer_T":
        return UformerConfig(img_size=img_size, embed_dim=16, modulator=True)
    if name == "Uformer_S":
        return UformerConfig(img_size=img_size, embed_dim=32, modulator=True)
    if name == "Uformer_B":
        return UformerConfig(img_size=img_size, embed_dim=32, depths=(1, 2, 8, 8, 2, 8, 8, 2, 1),
                             modulator=True, dd_in=dd_in)
    if name == "tiny":   # BASELINE.json configs[0]: embed_dim=16, depths [1]*9
        return UformerConfig(img_size=img_size, embed_dim=16, depths=(1,) * 9, modulator=True)
    if name == "tiny32":  # smallest head_dim-32 model, used by the GPU parity tests
        return UformerConfig(img_size=img_size, embed_dim=32, depths=(1, 2, 2, 2, 2, 2, 2, 2, 1), modulator=True)
    raise ValueError(f"unknown arch {name!r}")


# ---- state_dict layout ------------------------------------------------------------------
def block_spec(prefix: str, C: int, heads: int, win: int, modulator: bool,
               mlp_ratio: float = 4.0) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) of one LeWinTransformerBlock, in registration order."""
    hid = int(C * mlp_ratio)
    if modulator:
        yield prefix + "modulator.weight", (win * win, C), "embedding"
    yield prefix + "norm1.weight", (C,), "ln_w"
    yield prefix + "norm1.bias", (C,), "ln_b"
    yield prefix + "attn.relative_position_bias_table", ((2 * win - 1) ** 2, heads), "rpb"
    yield prefix + "attn.relative_position_index", (win * win, win * win), "rpi"
    yield prefix + "attn.qkv.to_q.weight", (C, C), "linear_w"
    yield prefix + "attn.qkv.to_q.bias", (C,), "linear_b"
    yield prefix + "attn.qkv.to_kv.weight", (2 * C, C), "linear_w"
    yield prefix + "attn.qkv.to_kv.bias", (2 * C,), "linear_b"
    yield prefix + "attn.proj.weight", (C, C), "linear_w"
    yield prefix + "attn.proj.bias", (C,), "linear_b"
    yield prefix + "norm2.weight", (C,), "ln_w"
    yield prefix + "norm2.bias", (C,), "ln_b"
    yield prefix + "mlp.linear1.0.weight", (hid, C), "linear_w"
    yield prefix + "mlp.linear1.0.bias", (hid,), "linear_b"
    yield prefix + "mlp.dwconv.0.weight", (hid, 1, 3, 3), "conv_w"
    yield prefix + "mlp.dwconv.0.bias", (hid,), "conv_b"
    yield prefix + "mlp.linear2.0.weight", (C, hid), "linear_w"
    yield prefix + "mlp.linear2.0.bias", (C,), "linear_b"


def state_dict_spec(cfg: UformerConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Every key of the reference ``Uformer.state_dict()`` with shape, in reference order."""
    e = cfg.embed_dim
    dims = cfg.stage_dims()
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    spec.append(("input_proj.proj.0.weight", (e, cfg.dd_in, 3, 3), "conv_w"))
    spec.append(("input_proj.proj.0.bias", (e,), "conv_b"))
    spec.append(("output_proj.proj.0.weight", (cfg.in_chans, 2 * e, 3, 3), "conv_w"))
    spec.append(("output_proj.proj.0.bias", (cfg.in_chans,), "conv_b"))

    def stage(s: int):
        for i in range(cfg.depths[s]):
            spec.extend(block_spec(f"{STAGES[s]}.blocks.{i}.", dims[s], cfg.num_heads[s],
                                   cfg.win_size, cfg.stage_has_modulator(s), cfg.mlp_ratio))

    for s in range(4):
        stage(s)
        spec.append((f"dowsample_{s}.conv.0.weight", (2 * dims[s], dims[s], 4, 4), "conv_w"))
        spec.append((f"dowsample_{s}.conv.0.bias", (2 * dims[s],), "conv_b"))
    stage(4)
    for k, (cin, cout) in enumerate(cfg.upsample_io()):
        spec.append((f"upsample_{k}.deconv.0.weight", (cin, cout, 2, 2), "deconv_w"))
        spec.append((f"upsample_{k}.deconv.0.bias", (cout,), "conv_b"))
        stage(5 + k)
    return spec


def relative_position_index(win: int) -> torch.Tensor:
    """(win², win²) int64 buffer, model.py:467-477: (yi-yj+win-1)*(2win-1) + (xi-xj+win-1)."""
    c = torch.arange(win)
    ys, xs = torch.meshgrid(c, c, indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    return ((ys[:, None] - ys[None, :] + win - 1) * (2 * win - 1)
            + (xs[:, None] - xs[None, :] + win - 1)).to(torch.int64)


def synth_state_dict(cfg: UformerConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Deterministic "trained-like" weights in the reference layout (set W1).

    Every tensor is drawn from its own CPU generator seeded by (seed, key), so the result
    does not depend on key order.  Biases, LN affine params and the relative-position
    tables are non-trivial so those code paths are exercised by the parity tests.
    """
    out: Dict[str, torch.Tensor] = {}
    for key, shape, kind in state_dict_spec(cfg):
        h = int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:6], "little")
        g = torch.Generator().manual_seed(h)
        if kind == "rpi":
            out[key] = relative_position_index(cfg.win_size)
            continue
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        if kind == "linear_w":
            t = 0.02 * n.clamp_(-2, 2) * 2.5          # a bit hotter than init so branches matter
        elif kind == "linear_b":
            t = 0.05 * n
        elif kind == "ln_w":
            t = 1.0 + 0.05 * n
        elif kind == "ln_b":
            t = 0.05 * n
        elif kind == "rpb":
            t = 0.2 * n
        elif kind == "embedding":
            t = 0.5 * n
        elif kind in ("conv_w", "deconv_w"):
            fan_in = shape[1] * shape[2] * shape[3] if kind == "conv_w" else shape[0] * shape[2] * shape[3]
            t = n.clamp_(-2, 2) * (0.6 / math.sqrt(fan_in))
        elif kind == "conv_b":
            t = 0.05 * n
        else:
            raise AssertionError(kind)
        out[key] = t.contiguous()
    return out


def synth_input(B: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """U[0,1) images, the range ``load_img`` produces (utils/image_utils.py:31-35)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, H, W, generator=g, dtype=torch.float32)
