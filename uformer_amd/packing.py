"""Weight re-layout from the reference ``state_dict`` to what the kernels read.

Pure tensor reshapes (no arithmetic except the dtype cast of GEMM weights to the operand
type); runs on whatever device the parameters live on.  Layouts are documented in
``include/uformer_hip.h``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from . import _lib

Tensor = torch.Tensor


def rpb_dense(table: Tensor, index: Tensor) -> Tensor:
    """(225,heads) table + (64,64) int64 index -> (heads,64,64) f32.  model.py:500-502."""
    n = index.shape[0]
    return table.detach()[index.reshape(-1)].reshape(n, n, -1).permute(2, 0, 1).contiguous().float()


def pack_frag(w: Tensor, dtype: torch.dtype) -> Tensor:
    """nn.Linear weight (N,K) -> fragment-major (N/16, ceil(K/32), 4, 16, 8) of ``dtype``: one wave-level
    MFMA-operand load then reads 1 KiB contiguous (layout formula in include/uformer_hip.h)."""
    N, K = w.shape
    KP = (K + 31) // 32 * 32
    w = w.detach().to(dtype)
    if KP != K:
        w = torch.nn.functional.pad(w, (0, KP - K))
    return w.reshape(N // 16, 16, KP // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def pack_rpb_table(dense: Tensor):
    """(heads,64,64) bias -> compact (heads,15,15) table T[h][dy+7][7-dx] with dy = yq-yk, dx = xq-xk, if the bias is
    Toeplitz in (dy,dx) (true for the reference's relative_position_index, model.py:467-477); else None."""
    h = dense.shape[0]
    c = torch.arange(8, device=dense.device)
    ys, xs = torch.meshgrid(c, c, indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    row = (ys[:, None] - ys[None, :] + 7)          # (64,64) dy+7
    pos = (7 - (xs[:, None] - xs[None, :]))        # (64,64) 7-dx
    tab = torch.zeros(h, 15, 15, dtype=torch.float32, device=dense.device)
    tab[:, row, pos] = dense.float()
    if not torch.equal(tab[:, row, pos], dense.float()):
        return None
    return tab.contiguous()


def pack_dwconv(w: Tensor) -> Tensor:
    """(C,1,3,3) -> (9,C) tap-major f32."""
    return w.detach().reshape(w.shape[0], 9).t().contiguous().float()


def pack_downsample(w: Tensor, dtype: torch.dtype) -> Tensor:
    """Conv2d weight (2C,C,4,4) -> (2C, 16C) with k = (ky*4+kx)*C + c."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(dtype)


def pack_upsample(w: Tensor, dtype: torch.dtype) -> Tensor:
    """ConvTranspose2d weight (Cin,Cout,2,2) -> (4*Cout, Cin) with n = (dy*2+dx)*Cout + co."""
    return w.detach().permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous().to(dtype)


def pack_input_proj(w: Tensor) -> Tensor:
    """Conv2d weight (E,Cin,3,3) -> (Cin*9, E) f32."""
    return w.detach().permute(1, 2, 3, 0).reshape(-1, w.shape[0]).contiguous().float()


def pack_output_proj(w: Tensor) -> Tensor:
    """Conv2d weight (3,C2,3,3) -> (3, 9, C2) f32."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]).contiguous().float()


def pack_block(sd: Dict[str, Tensor], prefix: str, heads: int, shift: int, dtype: torch.dtype):
    """Returns (BlockParams, keepalive list).  ``sd`` maps reference keys to tensors on the GPU."""
    # small f32 tensors are COPIED, never aliased: the packed block must stay valid when a parameter's storage is replaced
    # (p.data = ..., EMA swaps) until the owner notices the new (data_ptr, _version) key and repacks
    f = lambda k: sd[prefix + k].detach().float().clone().contiguous()  # noqa: E731
    t = lambda k: sd[prefix + k].detach().to(dtype, copy=True).contiguous()  # noqa: E731
    wqkv = torch.cat([sd[prefix + "attn.qkv.to_q.weight"].detach(), sd[prefix + "attn.qkv.to_kv.weight"].detach()], 0)
    dense = rpb_dense(sd[prefix + "attn.relative_position_bias_table"], sd[prefix + "attn.relative_position_index"])
    keep: Dict[str, Tensor] = {
        "norm1_w": f("norm1.weight"), "norm1_b": f("norm1.bias"),
        "rpb_dense": dense,
        "wqkv_fm": pack_frag(wqkv, dtype),
        "bqkv": torch.cat([sd[prefix + "attn.qkv.to_q.bias"].detach(), sd[prefix + "attn.qkv.to_kv.bias"].detach()], 0).contiguous().float(),
        "wproj": t("attn.proj.weight"), "wproj_fm": pack_frag(sd[prefix + "attn.proj.weight"], dtype), "bproj": f("attn.proj.bias"),
        "norm2_w": f("norm2.weight"), "norm2_b": f("norm2.bias"),
        "w1_fm": pack_frag(sd[prefix + "mlp.linear1.0.weight"], dtype), "b1": f("mlp.linear1.0.bias"),
        "wdw9": pack_dwconv(sd[prefix + "mlp.dwconv.0.weight"]), "bdw": f("mlp.dwconv.0.bias"),
        "w2_fm": pack_frag(sd[prefix + "mlp.linear2.0.weight"], dtype), "b2": f("mlp.linear2.0.bias"),
    }
    if (prefix + "modulator.weight") in sd:
        keep["modulator"] = f("modulator.weight")
    tab = pack_rpb_table(dense)
    if tab is not None:
        keep["rpb_tab"] = tab
    bp = _lib.BlockParams()
    for name, _ in _lib.BlockParams._fields_:
        if name in ("shift", "heads"):
            continue
        setattr(bp, name, keep[name].data_ptr() if name in keep else None)
    bp.shift = int(shift)
    bp.heads = int(heads)
    return bp, keep


class PackedModel:
    """All packed weights of one ``Uformer`` + the ``uf_model_desc`` that points at them."""

    def __init__(self, cfg, sd: Dict[str, Tensor], dtype: torch.dtype):
        from .spec import STAGES
        self.dtype = dtype
        self.keep: List[object] = []
        shifts = cfg.block_shifts()
        n_blocks = sum(cfg.depths)
        self.blocks = (_lib.BlockParams * n_blocks)()
        i = 0
        for s in range(9):
            for b in range(cfg.depths[s]):
                bp, keep = pack_block(sd, f"{STAGES[s]}.blocks.{b}.", cfg.num_heads[s], shifts[s][b], dtype)
                self.blocks[i] = bp
                self.keep.append(keep)
                i += 1
        d = _lib.ModelDesc()
        d.embed_dim, d.dd_in, d.in_chans = cfg.embed_dim, cfg.dd_in, cfg.in_chans
        for s in range(9):
            d.depths[s] = cfg.depths[s]
        d.blocks = C.cast(self.blocks, C.POINTER(_lib.BlockParams))
        g = {
            "in_w27": pack_input_proj(sd["input_proj.proj.0.weight"]),
            "in_b": sd["input_proj.proj.0.bias"].detach().float().clone().contiguous(),
            "out_w": pack_output_proj(sd["output_proj.proj.0.weight"]),
            "out_b": sd["output_proj.proj.0.bias"].detach().float().clone().contiguous(),
        }
        for name, tns in g.items():
            setattr(d, name, tns.data_ptr())
        self.keep.append(g)
        for k in range(4):
            dw = pack_downsample(sd[f"dowsample_{k}.conv.0.weight"], dtype)
            db = sd[f"dowsample_{k}.conv.0.bias"].detach().float().clone().contiguous()
            uw = pack_upsample(sd[f"upsample_{k}.deconv.0.weight"], dtype)
            ub = sd[f"upsample_{k}.deconv.0.bias"].detach().float().clone().contiguous()
            d.down_w[k], d.down_b[k], d.up_w[k], d.up_b[k] = dw.data_ptr(), db.data_ptr(), uw.data_ptr(), ub.data_ptr()
            # round 6: the Downsample weight also in the fragment-major layout its second form streams (2-byte operand types; N = 2C a multiple of 16)
            dwf = None
            if dtype in (torch.bfloat16, torch.float16) and dw.is_cuda and dw.shape[0] % 16 == 0 and dw.shape[1] % 32 == 0:
                from . import ops
                dwf = ops.pack_weight_fm(dw)
            d.down_w_fm[k] = dwf.data_ptr() if dwf is not None else None
            self.keep.append((dw, db, uw, ub, dwf))
        self.desc = d
