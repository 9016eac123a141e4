"""Thin torch-tensor wrappers over the C ABI (``include/uformer_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every wrapper extracts
raw pointers and calls ``libuformer_hip.so``.  No op has a CPU or eager fallback -- CPU tensors
raise.  Names follow the reference functions they replace (model.py:704-726 etc.).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import UF_BF16, UF_F16, UF_F32, UformerHipError

Tensor = torch.Tensor


def uf_dtype(dtype) -> int:
    if dtype in (UF_F32, UF_BF16, UF_F16) and not isinstance(dtype, torch.dtype):
        return int(dtype)
    if dtype == torch.float32:
        return UF_F32
    if dtype == torch.bfloat16:
        return UF_BF16
    if dtype == torch.float16:
        return UF_F16
    raise UformerHipError(f"unsupported operand dtype {dtype} (torch.float32, torch.bfloat16 or torch.float16)")


def torch_dtype(dt: int) -> torch.dtype:
    return {UF_BF16: torch.bfloat16, UF_F16: torch.float16}.get(dt, torch.float32)


def _dev(*ts: Tensor) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise UformerHipError("uformer_amd ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)
        if dev is not None and t.device != dev:
            raise UformerHipError("tensors on different devices")
        dev = t.device
    return dev


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _pack_frag(w: Tensor) -> Tensor:
    """Row-major nn.Linear weight -> the fragment-major layout the fused kernels stream (uf_pack_weight_fm)."""
    _dev(w)
    w = _c(w)
    N, K = w.shape
    lib = _lib.load()
    out = torch.empty(lib.uf_weight_fm_elems(N, K), dtype=w.dtype, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.uf_pack_weight_fm(_ptr(w), _ptr(out), N, K, uf_dtype(w.dtype), _stream()), "uf_pack_weight_fm")
    return out


def _c(t: Tensor, dtype: Optional[torch.dtype] = None) -> Tensor:
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------
# a1-a4 index ops
# ---------------------------------------------------------------------------------------
def window_partition(x: Tensor, win_size: int = 8, shift: int = 0) -> Tensor:
    """(B,H,W,C) -> (B*nW, 8, 8, C).  model.py:704-715; ``shift`` folds torch.roll (:957)."""
    if win_size != 8:
        raise UformerHipError("win_size must be 8")
    _dev(x)
    B, H, W, Cc = x.shape
    x = _c(x)
    if x.element_size() not in (2, 4):
        raise UformerHipError("window ops support 2- and 4-byte elements")
    out = torch.empty((B * (H // 8) * (W // 8), 8, 8, Cc), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_window_partition(_ptr(x), _ptr(out), B, H, W, Cc, shift, x.element_size(), _stream()),
                   "uf_window_partition")
    return out


def window_reverse(windows: Tensor, win_size: int, H: int, W: int, shift: int = 0) -> Tensor:
    """(B*nW, 8, 8, C) -> (B,H,W,C).  model.py:717-726; ``shift`` folds the roll back (:980)."""
    if win_size != 8:
        raise UformerHipError("win_size must be 8")
    _dev(windows)
    windows = _c(windows)
    Cc = windows.shape[-1]
    nW = (H // 8) * (W // 8)
    B = windows.shape[0] // nW
    out = torch.empty((B, H, W, Cc), dtype=windows.dtype, device=windows.device)
    with torch.cuda.device(windows.device):
        _lib.check(_lib.load().uf_window_reverse(_ptr(windows), _ptr(out), B, H, W, Cc, shift, windows.element_size(),
                                                 _stream()), "uf_window_reverse")
    return out


def shift_mask(H: int, W: int, shift: int, device) -> Tensor:
    """SW-MSA mask (nW,64,64) in {0,-100}.  model.py:924-942."""
    out = torch.empty(((H // 8) * (W // 8), 64, 64), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().uf_shift_mask(_ptr(out), H, W, shift, _stream()), "uf_shift_mask")
    return out


# ---------------------------------------------------------------------------------------
# float ops
# ---------------------------------------------------------------------------------------
def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, *, B: int, H: int, W: int, dtype, windowed: bool = False,
              shift: int = 0, modulator: Optional[Tensor] = None) -> Tensor:
    """x f32 (B*H*W, C) rows -> T (rows, C); optionally roll+partition+modulator (model.py:952-969)."""
    _dev(x, gamma, beta, modulator)
    dt = uf_dtype(dtype)
    x = _c(x, torch.float32)
    Cc = x.shape[-1]
    out = torch.empty((B * H * W, Cc), dtype=torch_dtype(dt), device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_layernorm_fwd(_ptr(x), Cc, _ptr(_c(gamma, torch.float32)), _ptr(_c(beta, torch.float32)),
                                                _ptr(None if modulator is None else _c(modulator, torch.float32)),
                                                _ptr(out), B, H, W, Cc, int(windowed), shift, dt, _stream()),
                   "uf_layernorm_fwd")
    return out


def linear(a: Tensor, w: Tensor, bias: Tensor, act: int = 0) -> Tensor:
    """out = act(a @ w.T + bias); a T(M,K), w T(N,K) -- nn.Linear (model.py:426-427,489,657,661)."""
    _dev(a, w, bias)
    dt = uf_dtype(a.dtype)
    a, w = _c(a), _c(w, a.dtype)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_linear_fwd(_ptr(a), _ptr(w), _ptr(_c(bias, torch.float32)), _ptr(out), M, N, K, act, dt,
                                             _stream()), "uf_linear_fwd")
    return out


def linear_residual(a: Tensor, w: Tensor, bias: Tensor, resid: Tensor, scale: Optional[Tensor], B: int, H: int, W: int, windowed: bool = False,
                    shift: int = 0) -> Tensor:
    """resid + scale[image] * (a @ w.T + bias) as f32 (B*H*W, N) raster rows; ``a`` in window-row order when ``windowed`` (window_reverse and
    the roll back happen in the store).  The projection / linear2 of a block with its residual add and DropPath (model.py:975-987)."""
    _dev(a, w, bias, resid)
    dt = uf_dtype(a.dtype)
    a, w, resid = _c(a), _c(w, a.dtype), _c(resid, torch.float32)
    M, K = a.shape
    N = w.shape[0]
    if M != B * H * W or resid.numel() != M * N:
        raise UformerHipError(f"linear_residual: a {tuple(a.shape)} / resid {tuple(resid.shape)} do not match B*H*W = {B * H * W}, N = {N}")
    sc = None if scale is None else _c(scale, torch.float32)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_linear_residual_fwd(_ptr(a), _ptr(w), _ptr(_c(bias, torch.float32)), _ptr(resid), _ptr(out), _ptr(sc) if sc is not None else None,
                                                      B, H, W, N, K, int(windowed), shift, dt, _stream()), "uf_linear_residual_fwd")
    return out


def qkv(a: Tensor, wqkv: Tensor, bqkv: Tensor, heads: int):
    """LinearProjection.forward (model.py:431-442) -> q (scaled), k, v^T per (window, head)."""
    _dev(a, wqkv, bqkv)
    dt = uf_dtype(a.dtype)
    a, wqkv = _c(a), _c(wqkv, a.dtype)
    M, Cc = a.shape
    hd = Cc // heads
    q = torch.empty((M // 64, heads, 64, hd), dtype=a.dtype, device=a.device)
    k = torch.empty_like(q)
    vt = torch.empty((M // 64, heads, hd, 64), dtype=a.dtype, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_qkv_fwd(_ptr(a), _ptr(wqkv), _ptr(_c(bqkv, torch.float32)), _ptr(q), _ptr(k), _ptr(vt), M,
                                          Cc, heads, dt, _stream()), "uf_qkv_fwd")
    return q, k, vt


def ln_qkv(x: Tensor, gamma: Tensor, beta: Tensor, wqkv: Tensor, bqkv: Tensor, heads: int, *, B: int, H: int, W: int,
           shift: int = 0, modulator: Optional[Tensor] = None):
    """Fused LN1 -> roll -> partition -> +modulator -> q,k,v^T (model.py:952-969, :431-442).  x f32 (B*H*W, C)."""
    _dev(x, gamma, beta, wqkv, bqkv, modulator)
    dt = uf_dtype(wqkv.dtype)
    x = _c(x, torch.float32)
    M, Cc = x.shape
    hd = Cc // heads
    q = torch.empty((M // 64, heads, 64, hd), dtype=wqkv.dtype, device=x.device)
    k = torch.empty_like(q)
    vt = torch.empty((M // 64, heads, hd, 64), dtype=wqkv.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_ln_qkv_fwd(_ptr(x), Cc, _ptr(_c(gamma, torch.float32)), _ptr(_c(beta, torch.float32)),
                                             _ptr(None if modulator is None else _c(modulator, torch.float32)), _ptr(_pack_frag(wqkv)),
                                             _ptr(_c(bqkv, torch.float32)), _ptr(q), _ptr(k), _ptr(vt), B, H, W, Cc, heads, shift,
                                             dt, _stream()), "uf_ln_qkv_fwd")
    return q, k, vt


def ln_linear_gelu(x: Tensor, gamma: Tensor, beta: Tensor, w1: Tensor, b1: Tensor) -> Tensor:
    """Fused LN2 -> linear1 -> GELU (model.py:987, :657-658).  x f32 (M, C); w1 T (N, C) -> T (M, N)."""
    _dev(x, gamma, beta, w1, b1)
    dt = uf_dtype(w1.dtype)
    x = _c(x, torch.float32)
    M, Cc = x.shape
    N = w1.shape[0]
    out = torch.empty((M, N), dtype=w1.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_ln_linear_gelu_fwd(_ptr(x), Cc, _ptr(_c(gamma, torch.float32)), _ptr(_c(beta, torch.float32)),
                                                     _ptr(_pack_frag(w1)), _ptr(_c(b1, torch.float32)), _ptr(out), M, N, Cc, dt, _stream()),
                   "uf_ln_linear_gelu_fwd")
    return out


def window_attention_core(q: Tensor, k: Tensor, vt: Tensor, bias_dense: Tensor, *, H: int, W: int, shift: int = 0,
                          mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T + bias + masks) v -> (n_windows*64, C).  model.py:498-519."""
    _dev(q, k, vt, bias_dense, mask)
    dt = uf_dtype(q.dtype)
    nwin, heads, _, hd = q.shape
    out = torch.empty((nwin * 64, heads * hd), dtype=q.dtype, device=q.device)
    if mask is not None:
        mask = _c(mask, torch.float32)
    with torch.cuda.device(q.device):
        _lib.check(_lib.load().uf_window_attention_fwd(_ptr(_c(q)), _ptr(_c(k)), _ptr(_c(vt)), _ptr(_c(bias_dense, torch.float32)),
                                                       _ptr(mask), 0 if mask is None else mask.shape[0], _ptr(out), nwin,
                                                       heads, hd, H, W, shift, dt, _stream()), "uf_window_attention_fwd")
    return out


def dwconv3x3_gelu(x: Tensor, w9: Tensor, bias: Tensor) -> Tensor:
    """x T(B,H,W,C) channel-last; w9 f32 (9,C); LeFF dwconv + GELU (model.py:659-660)."""
    _dev(x, w9, bias)
    dt = uf_dtype(x.dtype)
    x = _c(x)
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_dwconv3x3_gelu_fwd(_ptr(x), _ptr(_c(w9, torch.float32)), _ptr(_c(bias, torch.float32)),
                                                     _ptr(out), B, H, W, Cc, dt, _stream()), "uf_dwconv3x3_gelu_fwd")
    return out


def dwconv3x3(x: Tensor, w9: Tensor, bias: Optional[Tensor] = None, gelu: bool = False) -> Tensor:
    """The depthwise 3x3 stencil with optional bias / GELU.  With ``w9.flip(0)`` (taps flipped), no bias and no GELU it is
    the input gradient of the convolution."""
    _dev(x, w9)
    dt = uf_dtype(x.dtype)
    x = _c(x)
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    b = _c(bias, torch.float32) if bias is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_dwconv3x3_fwd(_ptr(x), _ptr(_c(w9, torch.float32)), _ptr(b) if b is not None else None, _ptr(out),
                                                B, H, W, Cc, 1 if gelu else 0, dt, _stream()), "uf_dwconv3x3_fwd")
    return out


def dwconv3x3_pre_gelu(x: Tensor, w9: Tensor, bias: Tensor, gelu_in: bool = False) -> Tuple[Tensor, Tensor]:
    """(pre, GELU(pre)) of the depthwise stencil in one pass (training forward keeps both).  ``gelu_in``: x is the PRE-activation of the GELU in
    front of the convolution and the kernel activates it as it loads it (uf_dwconv3x3_gelu_in_pre_gelu_fwd) -- same bits as passing the stored
    activation."""
    _dev(x, w9, bias)
    x = _c(x)
    B, H, W, Cc = x.shape
    pre, act = torch.empty_like(x), torch.empty_like(x)
    lib = _lib.load()
    fn, name = (lib.uf_dwconv3x3_gelu_in_pre_gelu_fwd, "uf_dwconv3x3_gelu_in_pre_gelu_fwd") if gelu_in else (lib.uf_dwconv3x3_pre_gelu_fwd, "uf_dwconv3x3_pre_gelu_fwd")
    with torch.cuda.device(x.device):
        _lib.check(fn(_ptr(x), _ptr(_c(w9, torch.float32)), _ptr(_c(bias, torch.float32)), _ptr(pre), _ptr(act), B, H, W, Cc, uf_dtype(x.dtype), _stream()), name)
    return pre, act


def dwconv3x3_mul_dgelu(dy: Tensor, w9_flipped: Tensor, pre: Tensor) -> Tensor:
    """Gradient through the depthwise conv and the GELU in front of it: stencil(dy; flipped taps) * GELU'(pre)."""
    _dev(dy, w9_flipped, pre)
    dy, pre = _c(dy), _c(pre, dy.dtype)
    B, H, W, Cc = dy.shape
    out = torch.empty_like(dy)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().uf_dwconv3x3_mul_dgelu(_ptr(dy), _ptr(_c(w9_flipped, torch.float32)), _ptr(pre), _ptr(out), B, H, W, Cc,
                                                      uf_dtype(dy.dtype), _stream()), "uf_dwconv3x3_mul_dgelu")
    return out


def linear_pre_gelu(a: Tensor, w: Tensor, bias: Tensor) -> Tuple[Tensor, Tensor]:
    """(a W^T + bias, GELU of it) in one pass; both T(M, N)."""
    _dev(a, w, bias)
    a, w = _c(a), _c(w, a.dtype)
    M, K = a.shape
    N = w.shape[0]
    pre = torch.empty((M, N), dtype=a.dtype, device=a.device)
    act = torch.empty_like(pre)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_linear_pre_gelu_fwd(_ptr(a), _ptr(w), _ptr(_c(bias, torch.float32)), _ptr(pre), _ptr(act), M, N, K, uf_dtype(a.dtype), _stream()),
                   "uf_linear_pre_gelu_fwd")
    return pre, act


def linear_mul_dgelu(dy: Tensor, w_t: Tensor, zero_bias: Tensor, pre: Tensor) -> Tensor:
    """(dy @ w_t.T) * GELU'(pre): input gradient of a Linear fed by a GELU; w_t (K_layer, N_layer) is the transposed weight."""
    _dev(dy, w_t, pre)
    dy, w_t, pre = _c(dy), _c(w_t, dy.dtype), _c(pre, dy.dtype)
    M, K = dy.shape
    N = w_t.shape[0]
    out = torch.empty((M, N), dtype=dy.dtype, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().uf_linear_mul_dgelu(_ptr(dy), _ptr(w_t), _ptr(_c(zero_bias, torch.float32)), _ptr(pre), _ptr(out), M, N, K, uf_dtype(dy.dtype), _stream()),
                   "uf_linear_mul_dgelu")
    return out


def gelu(a: Tensor) -> Tensor:
    """GELU(a) as a separate pass (nn.GELU, model.py:657-660); bf16 / f32."""
    _dev(a)
    dt = uf_dtype(a.dtype)
    a = _c(a)
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_gelu_fwd(_ptr(a), _ptr(out), a.numel(), dt, _stream()), "uf_gelu_fwd")
    return out


def gelu_bwd(a: Tensor, dy: Tensor) -> Tensor:
    """dy * GELU'(a), erf form (backward of nn.GELU, model.py:657-660).  a, dy: same shape and dtype (bf16 / f32)."""
    _dev(a, dy)
    dt = uf_dtype(a.dtype)
    a, dy = _c(a), _c(dy, a.dtype)
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().uf_gelu_bwd(_ptr(a), _ptr(dy), _ptr(out), a.numel(), dt, _stream()), "uf_gelu_bwd")
    return out


def layernorm_bwd(x: Tensor, gamma: Tensor, dy: Tensor):
    """Backward of nn.LayerNorm over the last dim of f32 rows: returns (dx, dgamma, dbeta).  model.py:881,888."""
    _dev(x, gamma, dy)
    Cc = x.shape[-1]
    x2, dy2 = _c(x, torch.float32).reshape(-1, Cc), _c(dy, torch.float32).reshape(-1, Cc)
    rows = x2.shape[0]
    dx = torch.empty_like(x2)
    dg = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_layernorm_bwd_workspace_bytes(rows, Cc)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_layernorm_bwd(_ptr(x2), Cc, _ptr(_c(gamma, torch.float32)), _ptr(dy2), Cc, _ptr(dx), Cc, _ptr(dg), _ptr(db),
                                        rows, Cc, _ptr(ws), nbytes, _stream()), "uf_layernorm_bwd")
    return dx.reshape(x.shape), dg, db


def layernorm_bwd_fused(x: Tensor, gamma: Tensor, dy: Tensor, B: int, H: int, W: int, add: Optional[Tensor] = None, windowed: bool = False, shift: int = 0,
                        cast: Optional[dict] = None):
    """layernorm_bwd reading dy (bf16 or f32 rows, in WINDOW order when ``windowed``) where the block backward has it, with an optional
    second gradient ``add`` (f32) summed into dx: returns (dx f32 (B*H*W, C) raster rows, dgamma, dbeta).  ``cast`` = dict(scale=(B,) f32 or
    None, windowed=bool, shift=int): a fourth result, (dx * scale[image]) in dy's dtype, in window order when cast['windowed'] -- the operand
    of the next GEMM of the backward (what ``grad_fork(dx, ...)`` returns), written by the same kernel (uf_layernorm_bwd_cast)."""
    _dev(x, gamma, dy)
    Cc = x.shape[-1]
    x2, dy2 = _c(x, torch.float32).reshape(-1, Cc), _c(dy).reshape(-1, Cc)
    add2 = None if add is None else _c(add, torch.float32).reshape(-1, Cc)
    dx = torch.empty_like(x2)
    dg = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_layernorm_bwd_workspace_bytes(x2.shape[0], Cc)
    ws = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        if cast is None:
            _lib.check(lib.uf_layernorm_bwd_fused(_ptr(x2), Cc, _ptr(_c(gamma, torch.float32)), _ptr(dy2), Cc, int(dy2.dtype == torch.float32), _ptr(add2), _ptr(dx), Cc,
                                                  _ptr(dg), _ptr(db), B, H, W, Cc, int(windowed), shift, uf_dtype(dy2.dtype), _ptr(ws), nbytes, _stream()),
                       "uf_layernorm_bwd_fused")
            return dx, dg, db
        out = torch.empty(B * H * W, Cc, dtype=dy2.dtype, device=x.device)
        scale = cast.get("scale")
        scale = None if scale is None else _c(scale, torch.float32)
        _lib.check(lib.uf_layernorm_bwd_cast(_ptr(x2), Cc, _ptr(_c(gamma, torch.float32)), _ptr(dy2), Cc, int(dy2.dtype == torch.float32), _ptr(add2), _ptr(dx), Cc,
                                             _ptr(dg), _ptr(db), B, H, W, Cc, int(windowed), shift, uf_dtype(dy2.dtype), _ptr(out), _ptr(scale),
                                             int(bool(cast.get("windowed", False))), int(cast.get("shift", 0)), _ptr(ws), nbytes, _stream()),
                   "uf_layernorm_bwd_cast")
    return dx, dg, db, out


def linear_wgrad(dy: Tensor, x: Tensor, with_bias: bool = True):
    """nn.Linear parameter gradients from the output gradient dy (..., N) and the layer input x (..., K), same dtype
    (bf16 / f32): returns (dW f32 (N,K), db f32 (N,) or None).  The input gradient is ``linear(dy, W.t())``."""
    _dev(dy, x)
    dt = uf_dtype(x.dtype)
    N, K = dy.shape[-1], x.shape[-1]
    dy2, x2 = _c(dy, x.dtype).reshape(-1, N), _c(x).reshape(-1, K)
    M = x2.shape[0]
    dW = torch.empty(N, K, dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if with_bias else None
    lib = _lib.load()
    nbytes = lib.uf_linear_wgrad_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_linear_wgrad(_ptr(dy2), N, _ptr(x2), K, _ptr(dW), _ptr(db) if db is not None else None, M, N, K, dt,
                                       _ptr(ws), nbytes, _stream()), "uf_linear_wgrad")
    return dW, db


def window_attention_bwd(q: Tensor, k: Tensor, vt: Tensor, bias: Tensor, do: Tensor, H: int, W: int, shift: int = 0,
                         mask: Optional[Tensor] = None):
    """Backward of window_attention_core: q (scaled), k (nW*heads,64,hd), vt (nW*heads,hd,64), bias f32 (heads,64,64),
    do (nW*64, heads*hd) -> (dq, dk, dvt, dbias f32 (heads,64,64) summed over windows).  model.py:494-519."""
    _dev(q, k, vt, bias, do)
    dt = uf_dtype(q.dtype)
    q, k, vt, do = _c(q), _c(k, q.dtype), _c(vt, q.dtype), _c(do, q.dtype)
    heads = bias.shape[0]
    hd = q.shape[-1]
    n_windows = q.numel() // (heads * 64 * hd)           # q may be (nW*heads, 64, hd) or (nW, heads, 64, hd)
    dq, dk, dvt = torch.empty_like(q), torch.empty_like(k), torch.empty_like(vt)
    dbias = torch.empty(heads, 64, 64, dtype=torch.float32, device=q.device)
    m = _c(mask, torch.float32) if mask is not None else None
    lib = _lib.load()
    nbytes = lib.uf_window_attention_bwd_workspace_bytes(n_windows, heads)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(lib.uf_window_attention_bwd(_ptr(q), _ptr(k), _ptr(vt), _ptr(_c(bias, torch.float32)), _ptr(m) if m is not None else None,
                                               m.shape[0] if m is not None else 0, _ptr(do), do.shape[-1], _ptr(dq), _ptr(dk), _ptr(dvt),
                                               _ptr(dbias), n_windows, heads, hd, H, W, shift, dt, _ptr(ws), nbytes, _stream()),
                   "uf_window_attention_bwd")
    return dq, dk, dvt, dbias


def window_attention_bwd_qkv(q: Tensor, k: Tensor, vt: Tensor, bias: Tensor, do: Tensor, H: int, W: int, shift: int = 0,
                             mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """window_attention_bwd writing the merged gradient of the q|k|v projection output: (dqkv (nW*64, 3C), dbias (heads,64,64))."""
    _dev(q, k, vt, bias, do)
    dt = uf_dtype(q.dtype)
    q, k, vt, do = _c(q), _c(k, q.dtype), _c(vt, q.dtype), _c(do, q.dtype)
    heads = bias.shape[0]
    hd = q.shape[-1]
    n_windows = q.numel() // (heads * 64 * hd)
    dqkv = torch.empty(n_windows * 64, 3 * heads * hd, dtype=q.dtype, device=q.device)
    dbias = torch.empty(heads, 64, 64, dtype=torch.float32, device=q.device)
    m = _c(mask, torch.float32) if mask is not None else None
    lib = _lib.load()
    nbytes = lib.uf_window_attention_bwd_workspace_bytes(n_windows, heads)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(lib.uf_window_attention_bwd_qkv(_ptr(q), _ptr(k), _ptr(vt), _ptr(_c(bias, torch.float32)), _ptr(m) if m is not None else None,
                                                   m.shape[0] if m is not None else 0, _ptr(do), do.shape[-1], _ptr(dqkv), _ptr(dbias), n_windows, heads, hd,
                                                   H, W, shift, dt, _ptr(ws), nbytes, _stream()), "uf_window_attention_bwd_qkv")
    return dqkv, dbias


def dwconv3x3_wgrad(h: Tensor, dc: Tensor):
    """Tap (9,C) and bias (C,) gradients of the depthwise 3x3 from its input h and output gradient dc, both T(B,H,W,C)."""
    _dev(h, dc)
    dt = uf_dtype(h.dtype)
    h, dc = _c(h), _c(dc, h.dtype)
    B, H, W, Cc = h.shape
    dw9 = torch.empty(9, Cc, dtype=torch.float32, device=h.device)
    db = torch.empty(Cc, dtype=torch.float32, device=h.device)
    lib = _lib.load()
    nbytes = lib.uf_dwconv3x3_wgrad_workspace_bytes(Cc, dt)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(lib.uf_dwconv3x3_wgrad(_ptr(h), _ptr(dc), _ptr(dw9), _ptr(db), B, H, W, Cc, dt, _ptr(ws), nbytes, _stream()),
                   "uf_dwconv3x3_wgrad")
    return dw9, db


def dwconv3x3_bwd(dc: Tensor, w9_flipped: Tensor, pre: Tensor):
    """Whole backward of the LeFF depthwise conv behind its GELU, one pass over dc (model.py:657-660): returns (da = stencil(dc; flipped
    taps) * GELU'(pre) -- as dwconv3x3_mul_dgelu --, dw9 (9,C), db (C,) -- as dwconv3x3_wgrad(GELU(pre), dc))."""
    _dev(dc, w9_flipped, pre)
    dt = uf_dtype(dc.dtype)
    dc, pre = _c(dc), _c(pre, dc.dtype)
    B, H, W, Cc = dc.shape
    da = torch.empty_like(dc)
    dw9 = torch.empty(9, Cc, dtype=torch.float32, device=dc.device)
    db = torch.empty(Cc, dtype=torch.float32, device=dc.device)
    lib = _lib.load()
    nbytes = lib.uf_dwconv3x3_bwd_workspace_bytes(Cc, dt)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dc.device)
    with torch.cuda.device(dc.device):
        _lib.check(lib.uf_dwconv3x3_bwd(_ptr(dc), _ptr(_c(w9_flipped, torch.float32)), _ptr(pre), _ptr(da), _ptr(dw9), _ptr(db), B, H, W, Cc, dt,
                                        _ptr(ws), nbytes, _stream()), "uf_dwconv3x3_bwd")
    return da, dw9, db


def dwconv_linear2(h1: Tensor, w9: Tensor, bdw: Tensor, w2: Tensor, b2: Tensor, x: Tensor) -> Tensor:
    """x + linear2(GELU(dwconv3x3(h1))) (model.py:674-683, :987).  h1 T(B,H,W,4C); x f32 (B*H*W, C); returns new x."""
    _dev(h1, w9, bdw, w2, b2, x)
    dt = uf_dtype(h1.dtype)
    h1 = _c(h1)
    B, H, W, hid = h1.shape
    out = _c(x, torch.float32).clone()
    Cc = out.shape[-1]
    with torch.cuda.device(h1.device):
        _lib.check(_lib.load().uf_dwconv_linear2_fwd(_ptr(h1), _ptr(_c(w9, torch.float32)), _ptr(_c(bdw, torch.float32)),
                                                     _ptr(_pack_frag(_c(w2, h1.dtype))), _ptr(_c(b2, torch.float32)), _ptr(out), Cc, B, H, W, Cc,
                                                     dt, _stream()), "uf_dwconv_linear2_fwd")
    return out


def lewin_attn_train_fwd(bp, x: Tensor, B: int, H: int, W: int, heads: int, dtype, drop_attn: Optional[Tensor] = None):
    """Training forward of the attention half + linear1 of a LeWin block in ONE launch (uf_lewin_attn_train_fwd; model.py:951-987, :657-658): the fused
    window kernel with side stores of what the backward reads.  bp: _lib.BlockParams (the fused pack); x f32 (B*H*W, C), untouched.
    Returns (x1 f32 (M, C), xn, q, k, vt, o, z, a1) in the layouts of layernorm(windowed) / qkv / window_attention_core / layernorm / linear."""
    _dev(x, drop_attn)
    x = _c(x, torch.float32)
    M, Cc = x.shape
    T, hd = torch_dtype(uf_dtype(dtype)), Cc // heads
    dev = x.device
    x1 = torch.empty_like(x)
    xn, o, z = (torch.empty((M, Cc), dtype=T, device=dev) for _ in range(3))
    q = torch.empty((M // 64, heads, 64, hd), dtype=T, device=dev)
    k = torch.empty_like(q)
    vt = torch.empty((M // 64, heads, hd, 64), dtype=T, device=dev)
    a1 = torch.empty((M, 4 * Cc), dtype=T, device=dev)
    da = None if drop_attn is None else _c(drop_attn, torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().uf_lewin_attn_train_fwd(bp, _ptr(x), Cc, _ptr(x1), Cc, B, H, W, Cc, _ptr(da), uf_dtype(dtype), _ptr(xn), _ptr(q), _ptr(k), _ptr(vt),
                                                       _ptr(o), _ptr(z), _ptr(a1), _stream()), "uf_lewin_attn_train_fwd")
    return x1, xn, q, k, vt, o, z, a1


def pack_weight_fm(w: Tensor) -> Tensor:
    """Row-major (N, K) weight of a 2- or 4-byte operand type -> the fragment-major layout of ``uf_pack_weight_fm`` (N a multiple of 16)."""
    return _pack_frag(w)


def downsample(x: Tensor, w_packed: Tensor, bias: Tensor, B: int, H: int, W: int, w_fm: Optional[Tensor] = None) -> Tensor:
    """x f32 (B*H*W, C) -> f32 (B*H/2*W/2, 2C).  Downsample.forward model.py:739-746.  ``w_fm``: ``pack_weight_fm(w_packed)`` if the caller has it
    (the LDS-patch form of the kernel then streams it; same bits either way)."""
    _dev(x, w_packed, bias)
    x = _c(x, torch.float32)
    Cc = x.shape[-1]
    out = torch.empty((B * (H // 2) * (W // 2), 2 * Cc), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if w_fm is not None:
            _lib.check(_lib.load().uf_downsample_fm_fwd(_ptr(x), Cc, _ptr(_c(w_packed)), _ptr(_c(w_fm)), _ptr(_c(bias, torch.float32)), _ptr(out),
                                                        2 * Cc, B, H, W, Cc, uf_dtype(w_packed.dtype), _stream()), "uf_downsample_fm_fwd")
        else:
            _lib.check(_lib.load().uf_downsample_fwd(_ptr(x), Cc, _ptr(_c(w_packed)), _ptr(_c(bias, torch.float32)), _ptr(out),
                                                     2 * Cc, B, H, W, Cc, uf_dtype(w_packed.dtype), _stream()), "uf_downsample_fwd")
    return out


def upsample(x: Tensor, w_packed: Tensor, bias: Tensor, B: int, H: int, W: int, out: Optional[Tensor] = None,
             ld_o: Optional[int] = None) -> Tensor:
    """x f32 (B*H*W, Cin) -> f32 (B*2H*2W, Cout).  Upsample.forward model.py:765-771."""
    _dev(x, w_packed, bias)
    x = _c(x, torch.float32)
    Cin = x.shape[-1]
    Cout = w_packed.shape[0] // 4
    if out is None:
        out = torch.empty((B * 4 * H * W, Cout), dtype=torch.float32, device=x.device)
        ld_o = Cout
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_upsample_fwd(_ptr(x), Cin, _ptr(_c(w_packed)), _ptr(_c(bias, torch.float32)), _ptr(out), ld_o,
                                               B, H, W, Cin, Cout, uf_dtype(w_packed.dtype), _stream()), "uf_upsample_fwd")
    return out


def input_proj(img: Tensor, w27: Tensor, bias: Tensor) -> Tensor:
    """img f32 NCHW -> tokens f32 (B*H*W, E).  InputProj.forward model.py:795-800."""
    _dev(img, w27, bias)
    img = _c(img, torch.float32)
    B, Cin, H, W = img.shape
    E = w27.shape[1]
    out = torch.empty((B * H * W, E), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().uf_input_proj_fwd(_ptr(img), _ptr(_c(w27, torch.float32)), _ptr(_c(bias, torch.float32)),
                                                 _ptr(out), E, B, Cin, H, W, E, _stream()), "uf_input_proj_fwd")
    return out


def output_proj(x: Tensor, w: Tensor, bias: Tensor, B: int, H: int, W: int, img: Optional[Tensor] = None) -> Tensor:
    """tokens f32 (B*H*W, C2) -> f32 NCHW (B,3,H,W) (+ img).  OutputProj.forward model.py:828-836, :1305."""
    _dev(x, w, bias, img)
    x = _c(x, torch.float32)
    C2 = x.shape[-1]
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    if img is not None:
        img = _c(img, torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_output_proj_fwd(_ptr(x), C2, _ptr(_c(w, torch.float32)), _ptr(_c(bias, torch.float32)),
                                                  _ptr(img), _ptr(out), B, H, W, C2, int(img is not None), _stream()),
                   "uf_output_proj_fwd")
    return out


# ---------------------------------------------------------------------------------------
# training-step tail, metrics, arbitrary-resolution wrapper, input pipeline
# ---------------------------------------------------------------------------------------
def _byref(struct):
    import ctypes
    return ctypes.byref(struct)


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def charbonnier(y: Tensor, target: Tensor, eps: float = 1e-3, with_grad: bool = True, grad_scale: float = 1.0):
    """CharbonnierLoss.forward (losses.py:41-52) and d loss / d y in ONE pass: returns (loss f32 0-dim tensor, dy or None)."""
    _dev(y, target)
    y, target = _c(y, torch.float32), _c(target, torch.float32)
    n = y.numel()
    loss = torch.empty(1, dtype=torch.float32, device=y.device)
    dy = torch.empty_like(y) if with_grad else None
    lib = _lib.load()
    nbytes = lib.uf_charbonnier_workspace_bytes(n)
    ws = _ws(nbytes, y.device)
    with torch.cuda.device(y.device):
        _lib.check(lib.uf_charbonnier_fwd_bwd(_ptr(y), _ptr(target), _ptr(dy), _ptr(loss), n, eps, grad_scale, _ptr(ws), nbytes, _stream()),
                   "uf_charbonnier_fwd_bwd")
    return loss.reshape(()), dy


def adamw_step(params, grads, exp_avg, exp_avg_sq, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.02,
               step: int = 1, grad_scale: float = 1.0) -> None:
    """One torch.optim.AdamW step over lists of f32 tensors, in place (train/train_denoise.py:77).  ``step`` counts from 1."""
    import ctypes as C
    n = len(params)
    if not (len(grads) == len(exp_avg) == len(exp_avg_sq) == n):
        raise UformerHipError("adamw_step: list lengths differ")
    if n == 0:
        return
    dev = _dev(*params, *grads, *exp_avg, *exp_avg_sq)
    for group in (params, grads, exp_avg, exp_avg_sq):
        for t in group:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise UformerHipError("adamw_step: tensors must be contiguous float32")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
    numel = (C.c_longlong * n)(*[p.numel() for p in params])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().uf_adamw_step(arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, n, lr, betas[0], betas[1], eps,
                                             weight_decay, int(step), grad_scale, _stream()), "uf_adamw_step")


def adamw_step_scaled(params, grads, exp_avg, exp_avg_sq, scaler_state: Tensor, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.02,
                      grad_scale: float = 1.0) -> None:
    """``adamw_step`` under a device-resident dynamic loss scale (uf_adamw_step_scaled): gradients x grad_scale / scale, nothing written when
    ``scaler_state[2]`` (found_inf) is set, bias corrections at step ``scaler_state[4] + 1``.  No host synchronisation."""
    import ctypes as C
    n = len(params)
    if not (len(grads) == len(exp_avg) == len(exp_avg_sq) == n):
        raise UformerHipError("adamw_step_scaled: list lengths differ")
    if n == 0:
        return
    dev = _dev(*params, *grads, *exp_avg, *exp_avg_sq, scaler_state)
    for group in (params, grads, exp_avg, exp_avg_sq, [scaler_state]):
        for t in group:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise UformerHipError("adamw_step_scaled: tensors must be contiguous float32")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
    numel = (C.c_longlong * n)(*[p.numel() for p in params])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().uf_adamw_step_scaled(arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, n, lr, betas[0], betas[1], eps,
                                                    weight_decay, grad_scale, _ptr(scaler_state), _stream()), "uf_adamw_step_scaled")


def grad_scaler_check(grads, scaler_state: Tensor) -> None:
    """scaler_state[2] = 1 if any gradient element is inf / nan (uf_grad_scaler_check)."""
    import ctypes as C
    grads = [g for g in grads if g is not None]
    n = len(grads)
    if n == 0:
        return
    dev = _dev(*grads, scaler_state)
    for t in grads:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise UformerHipError("grad_scaler_check: gradients must be contiguous float32")
    arr = (C.c_void_p * n)(*[t.data_ptr() for t in grads])
    numel = (C.c_longlong * n)(*[t.numel() for t in grads])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().uf_grad_scaler_check(arr, numel, n, _ptr(scaler_state), _stream()), "uf_grad_scaler_check")


def grad_scaler_update(scaler_state: Tensor, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000) -> None:
    """GradScaler.update() on the device (uf_grad_scaler_update)."""
    _dev(scaler_state)
    with torch.cuda.device(scaler_state.device):
        _lib.check(_lib.load().uf_grad_scaler_update(_ptr(scaler_state), growth_factor, backoff_factor, int(growth_interval), _stream()), "uf_grad_scaler_update")


def batch_mse(a: Tensor, b: Tensor, clamp01: bool = True) -> Tensor:
    """Per-image mean squared difference of (B,C,H,W) f32 images, clamped to [0,1] first (myPSNR, utils/image_utils.py:40-44)."""
    _dev(a, b)
    a, b = _c(a, torch.float32), _c(b, torch.float32)
    if a.shape != b.shape or a.dim() != 4:
        raise UformerHipError(f"batch_mse: shapes {tuple(a.shape)} / {tuple(b.shape)}")
    B, Cc, H, W = a.shape
    out = torch.empty(B, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    nbytes = lib.uf_image_metric_workspace_bytes(B, Cc, H, W)
    ws = _ws(nbytes, a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.uf_batch_mse(_ptr(a), _ptr(b), _ptr(out), B, Cc, H, W, int(clamp01), _ptr(ws), nbytes, _stream()), "uf_batch_mse")
    return out


def batch_ssim(a: Tensor, b: Tensor) -> Tensor:
    """Per-image SSIM of (B,C,H,W) f32 images in [0,1] as calculate_ssim computes it (utils/caculate_psnr_ssim.py:35-81)."""
    _dev(a, b)
    a, b = _c(a, torch.float32), _c(b, torch.float32)
    if a.shape != b.shape or a.dim() != 4:
        raise UformerHipError(f"batch_ssim: shapes {tuple(a.shape)} / {tuple(b.shape)}")
    B, Cc, H, W = a.shape
    out = torch.empty(B, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    nbytes = lib.uf_image_metric_workspace_bytes(B, Cc, H, W)
    ws = _ws(nbytes, a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.uf_batch_ssim(_ptr(a), _ptr(b), _ptr(out), B, Cc, H, W, _ptr(ws), nbytes, _stream()), "uf_batch_ssim")
    return out


def expand2square(img: Tensor, factor: float = 128.0, with_mask: bool = True):
    """test/test_sidd.py:79-92 for a batch: (B,C,h,w) -> zero canvas (B,C,X,X) with the image centred, + (B,1,X,X) mask."""
    import math
    _dev(img)
    img = _c(img, torch.float32)
    B, Cc, h, w = img.shape
    X = int(math.ceil(max(h, w) / float(factor)) * factor)
    canvas = torch.empty((B, Cc, X, X), dtype=torch.float32, device=img.device)
    mask = torch.empty((B, 1, X, X), dtype=torch.float32, device=img.device) if with_mask else None
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().uf_expand2square(_ptr(img), _ptr(canvas), _ptr(mask), B, Cc, h, w, X, _stream()), "uf_expand2square")
    return canvas, mask


def crop_clamp(canvas: Tensor, h: int, w: int, clamp: bool = True) -> Tensor:
    """masked_select crop of the restored canvas back to (B,C,h,w) + clamp to [0,1] (test/test_sidd.py:108-109)."""
    _dev(canvas)
    canvas = _c(canvas, torch.float32)
    B, Cc, X, X2 = canvas.shape
    if X != X2:
        raise UformerHipError("crop_clamp: square canvas expected")
    out = torch.empty((B, Cc, h, w), dtype=torch.float32, device=canvas.device)
    with torch.cuda.device(canvas.device):
        _lib.check(_lib.load().uf_crop_clamp(_ptr(canvas), _ptr(out), B, Cc, h, w, X, int(clamp), _stream()), "uf_crop_clamp")
    return out


def crop_augment(frames: Tensor, meta: Tensor, ps: int, hwc: bool = False) -> Tensor:
    """frames: (N,3,H,W) or (N,H,W,3) (hwc) uint8 / f32 on the GPU; meta int32 (B,4) = [frame index, r0, c0, transform 0..7]
    -> (B,3,ps,ps) f32 patches (dataset/dataset_denoise.py:54-70; uint8 is divided by 255 like load_img)."""
    _dev(frames, meta)
    if frames.dtype not in (torch.uint8, torch.float32):
        raise UformerHipError("crop_augment: frames must be uint8 or float32")
    frames, meta = _c(frames), _c(meta, torch.int32)
    N = frames.shape[0]
    H, W = (frames.shape[1], frames.shape[2]) if hwc else (frames.shape[2], frames.shape[3])
    B = meta.shape[0]
    out = torch.empty((B, 3, ps, ps), dtype=torch.float32, device=frames.device)
    with torch.cuda.device(frames.device):
        _lib.check(_lib.load().uf_crop_augment(_ptr(frames), int(frames.dtype == torch.uint8), int(hwc), _ptr(out), _ptr(meta), B, N, H, W, ps,
                                               _stream()), "uf_crop_augment")
    return out


def mixup(x: Tensor, lam: Tensor, perm: Tensor) -> Tensor:
    """lam[b] x[b] + (1 - lam[b]) x[perm[b]] (utils/dataset_utils.py:44-53)."""
    _dev(x, lam, perm)
    x = _c(x, torch.float32)
    B = x.shape[0]
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_mixup(_ptr(x), _ptr(out), _ptr(_c(lam.reshape(-1), torch.float32)), _ptr(_c(perm, torch.int32)), B,
                                        x.numel() // B, _stream()), "uf_mixup")
    return out


# ---------------------------------------------------------------------------------------
# a15: reductions / patch gathers of the backward, fused training forward of a block
# ---------------------------------------------------------------------------------------
def rows_sum(x: Tensor) -> Tensor:
    """f32 (N,) = sum over the rows of x (M, N) (bf16 / f32), fixed order (modulator gradient: x = d(xn) as (n_windows, 64*C))."""
    _dev(x)
    dt = uf_dtype(x.dtype)
    x = _c(x)
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_rows_sum_workspace_bytes(M, N)
    ws = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_rows_sum(_ptr(x), N, _ptr(out), M, N, dt, _ptr(ws), nbytes, _stream()), "uf_rows_sum")
    return out


def lewin_block_bwd_workspace_bytes(B: int, H: int, W: int, C: int, heads: int, dtype) -> int:
    return int(_lib.load().uf_lewin_block_bwd_workspace_bytes(B, H, W, C, heads, uf_dtype(dtype)))


def _block_grads(C: int, heads: int, modulator: bool, device):
    """One flat f32 buffer holding all parameter gradients of a block, its uf_block_grads view and {field: tensor view}."""
    shapes = [("norm1_w", (C,)), ("norm1_b", (C,)), ("norm2_w", (C,)), ("norm2_b", (C,)), ("modulator", (64, C) if modulator else None),
              ("rpb_table", (225, heads)), ("wqkv", (3 * C, C)), ("bqkv", (3 * C,)), ("wproj", (C, C)), ("bproj", (C,)), ("w1", (4 * C, C)), ("b1", (4 * C,)),
              ("wdw", (4 * C, 1, 3, 3)), ("bdw", (4 * C,)), ("w2", (C, 4 * C)), ("b2", (C,))]
    total = 0
    offs = {}
    for name, shp in shapes:
        if shp is None:
            continue
        n = 1
        for d in shp:
            n *= d
        offs[name] = (total, n, shp)
        total += (n + 63) // 64 * 64                          # 256-byte aligned pieces
    flat = torch.empty(total, dtype=torch.float32, device=device)
    views = {name: flat[o:o + n].view(shp) for name, (o, n, shp) in offs.items()}
    g = _lib.BlockGrads()
    for name, _ in shapes:
        setattr(g, name, views[name].data_ptr() if name in views else None)
    return flat, g, views


def lewin_block_bwd(tp, x: Tensor, dy: Tensor, drop_attn: Optional[Tensor], drop_leff: Optional[Tensor], B: int, H: int, W: int, heads: int, dtype,
                    ws: Optional[Tensor] = None, half: str = "block"):
    """Recomputation + backward of one LeWin block through uf_lewin_block_bwd (``half``: "block", or "leff" / "attn" for the two
    halves: x is then x1 resp. the block input and dy the gradient of that half's output).  tp: _lib.BlockTrainParams.
    Returns (dx f32 (B*H*W, C), {uf_block_grads field: f32 gradient tensor})."""
    _dev(x, dy)
    x, dy = _c(x, torch.float32), _c(dy, torch.float32)
    C = x.shape[-1]
    lib = _lib.load()
    dt = uf_dtype(dtype)
    nbytes = lib.uf_lewin_block_bwd_workspace_bytes(B, H, W, C, heads, dt)
    if ws is None or ws.numel() < nbytes:
        ws = _ws(nbytes, x.device)
    _flat, g, views = _block_grads(C, heads, bool(tp.modulator), x.device)
    dx = torch.empty_like(x)
    da = None if drop_attn is None else _c(drop_attn, torch.float32)
    dl = None if drop_leff is None else _c(drop_leff, torch.float32)
    with torch.cuda.device(x.device):
        if half == "block":
            rc = lib.uf_lewin_block_bwd(_byref(tp), _ptr(x), _ptr(dy), _ptr(dx), _ptr(da), _ptr(dl), _byref(g), B, H, W, C, dt, _ptr(ws), ws.numel(), _stream())
        elif half == "leff":
            rc = lib.uf_leff_bwd(_byref(tp), _ptr(x), _ptr(dy), _ptr(dx), _ptr(dl), _byref(g), B, H, W, C, dt, _ptr(ws), ws.numel(), _stream())
        else:
            rc = lib.uf_lewin_attn_bwd(_byref(tp), _ptr(x), _ptr(dy), _ptr(dx), _ptr(da), _byref(g), B, H, W, C, dt, _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "uf_lewin_block_bwd" if half == "block" else ("uf_leff_bwd" if half == "leff" else "uf_lewin_attn_bwd"))
    if half != "block":       # a half writes only its own parameters' gradients: never hand out the other half's uninitialised views
        mine = {"leff": ("norm2_w", "norm2_b", "w1", "b1", "wdw", "bdw", "w2", "b2"),
                "attn": ("norm1_w", "norm1_b", "modulator", "rpb_table", "wqkv", "bqkv", "wproj", "bproj")}[half]
        views = {k: v for k, v in views.items() if k in mine}
    return dx, views


def downsample_bwd(x: Tensor, dy: Tensor, w_pk_t: Tensor, B: int, H: int, W: int, add_to: Optional[Tensor] = None):
    """Backward of Downsample (Conv2d k4 s2 p1): x f32 rows (B*H*W, Cin) (row stride may exceed Cin), dy f32 (B*H/2*W/2, Cout),
    w_pk_t T (16 Cin, Cout).  Returns (dx -- ``add_to`` accumulated in place when given --, dW_pk f32 (Cout, 16 Cin) packed order, db)."""
    _dev(x, dy, w_pk_t)
    dy = _c(dy, torch.float32)
    Cin, Cout = x.shape[1], dy.shape[1]
    if x.stride(1) != 1:
        x = x.contiguous()
    dt = uf_dtype(w_pk_t.dtype)
    dx = add_to if add_to is not None else torch.empty(B * H * W, Cin, dtype=torch.float32, device=x.device)
    dW = torch.empty(Cout, 16 * Cin, dtype=torch.float32, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_downsample_bwd_workspace_bytes(B, H, W, Cin, Cout, dt)
    ws = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_downsample_bwd(_ptr(x), x.stride(0), _ptr(dy), _ptr(_c(w_pk_t)), _ptr(dx), dx.stride(0), int(add_to is not None), _ptr(dW), _ptr(db),
                                         B, H, W, Cin, Cout, dt, _ptr(ws), nbytes, _stream()), "uf_downsample_bwd")
    return dx, dW, db


def upsample_cat_bwd(d: Tensor, x: Tensor, w_pk_t: Tensor, B: int, H: int, W: int):
    """Backward of Upsample (ConvTranspose2d k2 s2) from the gradient ``d`` (B*2H*2W, ld) of the concat buffer whose first Cout columns
    it wrote; x f32 (B*H*W, Cin); w_pk_t T (Cin, 4 Cout).  Returns (dx f32 (B*H*W, Cin), dW_pk f32 (4 Cout, Cin) packed order, db)."""
    _dev(d, x, w_pk_t)
    x = _c(x, torch.float32)
    if d.stride(1) != 1 or d.dtype != torch.float32:
        d = d.float().contiguous()
    Cin, Cout = x.shape[1], w_pk_t.shape[1] // 4
    dt = uf_dtype(w_pk_t.dtype)
    dx = torch.empty(B * H * W, Cin, dtype=torch.float32, device=x.device)
    dW = torch.empty(4 * Cout, Cin, dtype=torch.float32, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_upsample_cat_bwd_workspace_bytes(B, H, W, Cin, Cout, dt)
    ws = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_upsample_cat_bwd(_ptr(d), d.stride(0), _ptr(x), Cin, _ptr(_c(w_pk_t)), _ptr(dx), _ptr(dW), _ptr(db), B, H, W, Cin, Cout, dt,
                                           _ptr(ws), nbytes, _stream()), "uf_upsample_cat_bwd")
    return dx, dW, db


def conv3x3_bwd(x: Tensor, dy_rows: Tensor, w: Tensor, B: int, H: int, W: int, nchw: bool = False, act_out: Optional[Tensor] = None, slope: float = 0.01,
                need_dx: bool = True) -> Tuple[Optional[Tensor], Tensor, Tensor]:
    """(dx, dW, db) of a 3x3 stride-1 pad-1 Conv2d with <= 4 channels on one side (InputProj / OutputProj): x = f32 token rows
    (B*H*W, Cin) or the NCHW image when ``nchw``; dy_rows f32 (B*H*W, Cout); ``act_out``: the stored LeakyReLU output rows (InputProj)."""
    _dev(x, dy_rows, w)
    x, dy_rows, w = _c(x, torch.float32), _c(dy_rows, torch.float32), _c(w, torch.float32)
    Cout, Cin = w.shape[0], w.shape[1]
    act_out = None if act_out is None else _c(act_out, torch.float32)
    dx = torch.empty_like(x) if need_dx else None
    dW, db = torch.empty_like(w), torch.empty(Cout, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.uf_conv3x3_bwd_workspace_bytes(B, H, W, Cin, Cout)
    ws = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.uf_conv3x3_bwd(_ptr(x), int(nchw), _ptr(dy_rows), _ptr(act_out), slope, _ptr(w), _ptr(dx), _ptr(dW), _ptr(db), B, H, W, Cin, Cout,
                                      _ptr(ws), nbytes, _stream()), "uf_conv3x3_bwd")
    return dx, dW, db


def residual_combine(a: Optional[Tensor], b: Tensor, scale: Optional[Tensor], B: int, H: int, W: int, windowed: bool = False, shift: int = 0) -> Tensor:
    """f32 (B*H*W, C) = (a or 0) + scale[image] * b; b (bf16/f32 rows) is in WINDOW order when ``windowed`` (window_reverse + roll
    back folded in), ``scale`` = f32 (B,) DropPath scales or None."""
    _dev(b)
    b = _c(b)
    C = b.shape[-1]
    out = torch.empty(B * H * W, C, dtype=torch.float32, device=b.device)
    a = None if a is None else _c(a, torch.float32)
    scale = None if scale is None else _c(scale, torch.float32)
    with torch.cuda.device(b.device):
        _lib.check(_lib.load().uf_residual_combine(_ptr(a), _ptr(b), int(b.dtype == torch.float32), _ptr(out), _ptr(scale), B, H, W, C, int(windowed), shift,
                                                   uf_dtype(b.dtype), _stream()), "uf_residual_combine")
    return out


def grad_fork(g1: Tensor, g2: Optional[Tensor], scale: Optional[Tensor], B: int, H: int, W: int, dtype, windowed: bool = False, shift: int = 0,
              want_sum: bool = False) -> Tuple[Optional[Tensor], Tensor]:
    """t = g1 (+ g2) (f32 raster rows); returns (t if ``want_sum`` else None, (t * scale[image]) cast to ``dtype``, in window order
    when ``windowed``)."""
    _dev(g1)
    g1 = _c(g1, torch.float32)
    g2 = None if g2 is None else _c(g2, torch.float32)
    C = g1.shape[-1]
    scale = None if scale is None else _c(scale, torch.float32)
    tot = torch.empty_like(g1) if want_sum else None
    out = torch.empty(B * H * W, C, dtype=dtype, device=g1.device)
    with torch.cuda.device(g1.device):
        _lib.check(_lib.load().uf_grad_fork(_ptr(g1), _ptr(g2), _ptr(tot), _ptr(out), _ptr(scale), B, H, W, C, int(windowed), shift, uf_dtype(dtype), _stream()),
                   "uf_grad_fork")
    return tot, out


def qkv_grad_merge(dq: Tensor, dk: Tensor, dvt: Tensor, heads: int) -> Tensor:
    """(n_windows*64, 3C) gradient of the fused q|k|v projection output from uf_window_attention_bwd's per-head tensors."""
    _dev(dq)
    dq, dk, dvt = _c(dq), _c(dk), _c(dvt)
    hd = dq.shape[-1]
    nW = dq.numel() // (heads * 64 * hd)
    out = torch.empty(nW * 64, 3 * heads * hd, dtype=dq.dtype, device=dq.device)
    with torch.cuda.device(dq.device):
        _lib.check(_lib.load().uf_qkv_grad_merge(_ptr(dq), _ptr(dk), _ptr(dvt), _ptr(out), nW, heads, hd, uf_dtype(dq.dtype), _stream()), "uf_qkv_grad_merge")
    return out


def rpb_table_grad(dbias: Tensor) -> Tensor:
    """(heads,64,64) dense bias gradient -> (225, heads) relative_position_bias_table gradient (model.py:500-502 transposed)."""
    _dev(dbias)
    dbias = _c(dbias, torch.float32)
    heads = dbias.shape[0]
    out = torch.empty(225, heads, dtype=torch.float32, device=dbias.device)
    with torch.cuda.device(dbias.device):
        _lib.check(_lib.load().uf_rpb_table_grad(_ptr(dbias), _ptr(out), heads, _stream()), "uf_rpb_table_grad")
    return out


def im2col(x: Tensor, B: int, H: int, W: int, Cin: int, k: int, stride: int, pad: int, dtype, nchw: bool = False) -> Tensor:
    """Patch matrix (B*Ho*Wo, ldc) of a conv input: f32 token rows (B*H*W, Cin) or an NCHW image; column (ky*k+kx)*Cin + c,
    ldc = k*k*Cin rounded up to 8 (zero columns)."""
    _dev(x)
    x = _c(x, torch.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    ldc = (k * k * Cin + 7) // 8 * 8
    dt = uf_dtype(dtype)
    cols = torch.empty((B * Ho * Wo, ldc), dtype=torch_dtype(dt), device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().uf_im2col(_ptr(x), Cin, _ptr(cols), ldc, B, H, W, Cin, k, stride, pad, int(nchw), dt, _stream()), "uf_im2col")
    return cols


def col2im(dcols: Tensor, B: int, H: int, W: int, Cin: int, k: int, stride: int, pad: int, nchw: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Transpose of im2col: f32 (B*H*W, Cin) rows (or NCHW) = gathered sum of the matching dcols entries; ``out`` given: added to it."""
    _dev(dcols, out)
    dt = uf_dtype(dcols.dtype)
    dcols = _c(dcols)
    acc = out is not None
    if out is None:
        out = torch.empty((B, Cin, H, W) if nchw else (B * H * W, Cin), dtype=torch.float32, device=dcols.device)
    elif not out.is_contiguous() or out.dtype != torch.float32:
        raise UformerHipError("col2im: out must be contiguous float32")
    with torch.cuda.device(dcols.device):
        _lib.check(_lib.load().uf_col2im(_ptr(dcols), dcols.shape[1], _ptr(out), Cin, B, H, W, Cin, k, stride, pad, int(nchw), int(acc), dt, _stream()),
                   "uf_col2im")
    return out


def lewin_block_train_fwd(bp, x: Tensor, B: int, H: int, W: int, dtype, drop_attn: Optional[Tensor] = None, drop_leff: Optional[Tensor] = None) -> Tensor:
    """Training forward of one LeWin block on the two fused inference kernels with DropPath scales (f32 (B,) per branch or None):
    returns the new stream; ``x`` (f32 (B*H*W, C)) is left untouched -- it is what the backward keeps."""
    _dev(x, drop_attn, drop_leff)
    dt = uf_dtype(dtype)
    y = _c(x, torch.float32).clone()
    Cc = y.shape[-1]
    lib = _lib.load()
    nbytes = lib.uf_block_workspace_bytes(B * H * W, Cc, dt)
    ws = _ws(nbytes, y.device)
    da = None if drop_attn is None else _c(drop_attn, torch.float32)
    dl = None if drop_leff is None else _c(drop_leff, torch.float32)
    with torch.cuda.device(y.device):
        _lib.check(lib.uf_lewin_block_train_fwd(bp, _ptr(y), Cc, B, H, W, Cc, _ptr(da), _ptr(dl), dt, _ptr(ws), nbytes, _stream()),
                   "uf_lewin_block_train_fwd")
    return y
