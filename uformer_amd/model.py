"""Drop-in ``Uformer(nn.Module)`` whose hot path runs in ``libuformer_hip.so`` (gfx950).

Boundary kept identical to the reference:
  * constructor kwargs of ``Uformer``                         model.py:1070-1077
  * ``forward(x, mask=None)`` on (B, dd_in, H, W) float       model.py:1269-1305
  * ``state_dict()`` key names / shapes / dtypes / order      SURVEY.md Appendix C
so reference checkpoints (with or without the ``module.`` prefix) load with ``strict=True``.

The sub-modules below carry the reference's names and parameter layout; their ``forward``
methods call the C ABI through ``uformer_amd.ops``.  ``torch.nn`` containers are used only to
HOLD parameters (so init, ``state_dict`` and optimizers behave as in the reference): no ATen
math runs on the hot path, and there is no CPU / eager fallback -- CPU inputs raise.

``compute_dtype`` selects the GEMM operand type: ``torch.bfloat16`` (MFMA bf16, f32 accumulate; BASELINE's headline type),
``torch.float16`` (IEEE half operands, the reference's own AMP type, train/train_denoise.py:180-184: the same MFMA rate and
within the 1e-3 output tolerance; train it under a loss scale as the reference does) or ``torch.float32`` (exact-f32 MFMA).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops, packing
from ._lib import UformerHipError
from .spec import STAGES, UformerConfig, relative_position_index

Tensor = torch.Tensor


def window_partition(x: Tensor, win_size: int, dilation_rate: int = 1) -> Tensor:
    """model.py:704-715 (dilated branch is dead code in the reference and not provided)."""
    if dilation_rate != 1:
        raise NotImplementedError("dilation_rate != 1 is never exercised by the reference (SURVEY.md section 2 row 1)")
    return ops.window_partition(x, win_size, 0)


def window_reverse(windows: Tensor, win_size: int, H: int, W: int, dilation_rate: int = 1) -> Tensor:
    """model.py:717-726."""
    if dilation_rate != 1:
        raise NotImplementedError("dilation_rate != 1 is never exercised by the reference")
    return ops.window_reverse(windows, win_size, H, W, 0)


def _to_compute(t: Tensor, dtype: torch.dtype) -> Tensor:
    return t.detach().to(dtype).contiguous()


class LinearProjection(nn.Module):
    """model.py:421-447.  Parameters ``to_q`` (C,C) and ``to_kv`` (2C,C)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., bias=True):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner_dim, bias=bias)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=bias)
        self.dim = dim
        self.inner_dim = inner_dim

    def packed(self, dtype):
        w = torch.cat([self.to_q.weight.detach(), self.to_kv.weight.detach()], 0).to(dtype).contiguous()
        b = torch.cat([self.to_q.bias.detach(), self.to_kv.bias.detach()], 0).float().contiguous()
        return w, b

    def flops(self, q_L, kv_L=None):
        kv_L = kv_L or q_L
        return q_L * self.dim * self.inner_dim + kv_L * self.dim * self.inner_dim * 2


class WindowAttention(nn.Module):
    """model.py:452-546 with ``token_projection='linear'`` (the only one any arch selects)."""

    def __init__(self, dim, win_size, num_heads, token_projection='linear', qkv_bias=True, qk_scale=None,
                 attn_drop=0., proj_drop=0.):
        super().__init__()
        if token_projection != 'linear':
            raise NotImplementedError("only token_projection='linear' is on the hot path (utils/model_utils.py:65-78)")
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale / attention dropout are never set by the reference archs")
        self.dim = dim
        self.win_size = tuple(win_size)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * self.win_size[0] - 1) * (2 * self.win_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(self.win_size[0]))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads, bias=qkv_bias)
        self.token_projection = token_projection
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: Tensor, attn_kv=None, mask: Optional[Tensor] = None, *, H: Optional[int] = None,
                W: Optional[int] = None, shift: int = 0, compute_dtype=torch.float32) -> Tensor:
        """x: (B_, 64, C) window tokens.  ``mask`` (nW,64,64) additive, as in model.py:508-512."""
        if attn_kv is not None:
            raise NotImplementedError("attn_kv (cross-modulator) is out of scope: never enabled by get_arch")
        B_, N, Cc = x.shape
        if N != 64:
            raise UformerHipError("window attention needs 8x8 = 64 tokens per window")
        if H is None or W is None:  # any geometry with the right window count serves a dense mask
            H, W = 8, 8 * B_
            if shift:
                raise UformerHipError("analytic shift mask needs H and W")
        a = _to_compute(x.reshape(B_ * N, Cc), compute_dtype)
        wqkv, bqkv = self.qkv.packed(compute_dtype)
        q, k, vt = ops.qkv(a, wqkv, bqkv, self.num_heads)
        bias = packing.rpb_dense(self.relative_position_bias_table, self.relative_position_index)
        o = ops.window_attention_core(q, k, vt, bias, H=H, W=W, shift=shift, mask=mask)
        y = ops.linear(o, _to_compute(self.proj.weight, compute_dtype), self.proj.bias.detach().float(), 0)
        return y.reshape(B_, N, Cc).to(x.dtype)

    def extra_repr(self) -> str:
        return f'dim={self.dim}, win_size={self.win_size}, num_heads={self.num_heads}'

    def flops(self, H, W):
        N = self.win_size[0] * self.win_size[1]
        nW = H * W / N
        flops = self.qkv.flops(H * W, H * W)
        flops += nW * self.num_heads * N * (self.dim // self.num_heads) * N * 2
        flops += nW * N * self.dim * self.dim
        return flops


class LeFF(nn.Module):
    """model.py:654-699."""

    def __init__(self, dim=32, hidden_dim=128, act_layer=nn.GELU, drop=0., use_eca=False):
        super().__init__()
        if use_eca or act_layer is not nn.GELU:
            raise NotImplementedError("LeFF is built with GELU and without ECA by every reference arch")
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim), act_layer())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1),
                                    act_layer())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))
        self.dim = dim
        self.hidden_dim = hidden_dim

    def forward(self, x: Tensor, compute_dtype=torch.float32) -> Tensor:
        bs, hw, c = x.shape
        hh = int(math.sqrt(hw))
        a = _to_compute(x.reshape(bs * hw, c), compute_dtype)
        h1 = ops.linear(a, _to_compute(self.linear1[0].weight, compute_dtype), self.linear1[0].bias.detach().float(), 1)
        h2 = ops.dwconv3x3_gelu(h1.reshape(bs, hh, hh, self.hidden_dim), packing.pack_dwconv(self.dwconv[0].weight),
                                self.dwconv[0].bias.detach().float())
        y = ops.linear(h2.reshape(bs * hw, self.hidden_dim), _to_compute(self.linear2[0].weight, compute_dtype),
                       self.linear2[0].bias.detach().float(), 0)
        return y.reshape(bs, hw, c).to(x.dtype)

    def flops(self, H, W):
        return H * W * self.dim * self.hidden_dim + H * W * self.hidden_dim * 9 + H * W * self.hidden_dim * self.dim


class Downsample(nn.Module):
    """model.py:730-753."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=4, stride=2, padding=1))
        self.in_channel = in_channel
        self.out_channel = out_channel

    def forward(self, x: Tensor, compute_dtype=torch.float32) -> Tensor:
        B, L, Cc = x.shape
        H = W = int(math.sqrt(L))
        y = ops.downsample(x.reshape(B * L, Cc), packing.pack_downsample(self.conv[0].weight, compute_dtype),
                           self.conv[0].bias.detach().float(), B, H, W)
        return y.reshape(B, L // 4, self.out_channel)

    def flops(self, H, W):
        return H / 2 * W / 2 * self.in_channel * self.out_channel * 4 * 4


class Upsample(nn.Module):
    """model.py:756-778."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.deconv = nn.Sequential(nn.ConvTranspose2d(in_channel, out_channel, kernel_size=2, stride=2))
        self.in_channel = in_channel
        self.out_channel = out_channel

    def forward(self, x: Tensor, compute_dtype=torch.float32) -> Tensor:
        B, L, Cc = x.shape
        H = W = int(math.sqrt(L))
        y = ops.upsample(x.reshape(B * L, Cc), packing.pack_upsample(self.deconv[0].weight, compute_dtype),
                         self.deconv[0].bias.detach().float(), B, H, W)
        return y.reshape(B, 4 * L, self.out_channel)

    def flops(self, H, W):
        return H * W * self.in_channel * self.out_channel   # exact (the reference over-counts 4x, model.py:776)


class InputProj(nn.Module):
    """model.py:781-811."""

    def __init__(self, in_channel=3, out_channel=64, kernel_size=3, stride=1, norm_layer=None, act_layer=nn.LeakyReLU):
        super().__init__()
        if norm_layer is not None or kernel_size != 3 or stride != 1 or act_layer is not nn.LeakyReLU:
            raise NotImplementedError("InputProj: only the configuration Uformer builds (model.py:1100)")
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2),
                                  act_layer(inplace=True))
        self.in_channel = in_channel
        self.out_channel = out_channel

    def forward(self, x: Tensor) -> Tensor:
        B, Cc, H, W = x.shape
        y = ops.input_proj(x, packing.pack_input_proj(self.proj[0].weight), self.proj[0].bias.detach().float())
        return y.reshape(B, H * W, self.out_channel)

    def flops(self, H, W):
        return H * W * self.in_channel * self.out_channel * 9


class OutputProj(nn.Module):
    """model.py:814-846."""

    def __init__(self, in_channel=64, out_channel=3, kernel_size=3, stride=1, norm_layer=None, act_layer=None):
        super().__init__()
        if norm_layer is not None or act_layer is not None or kernel_size != 3 or stride != 1 or out_channel != 3:
            raise NotImplementedError("OutputProj: only the configuration Uformer builds (model.py:1101)")
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2))
        self.in_channel = in_channel
        self.out_channel = out_channel

    def forward(self, x: Tensor, img: Optional[Tensor] = None) -> Tensor:
        B, L, Cc = x.shape
        H = W = int(math.sqrt(L))
        return ops.output_proj(x.reshape(B * L, Cc), packing.pack_output_proj(self.proj[0].weight),
                               self.proj[0].bias.detach().float(), B, H, W, img)

    def flops(self, H, W):
        return H * W * self.in_channel * self.out_channel * 9


class LeWinTransformerBlock(nn.Module):
    """model.py:850-1008 (``token_mlp='leff'``, no cross-modulator)."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 token_projection='linear', token_mlp='leff', modulator=False, cross_modulator=False):
        super().__init__()
        if cross_modulator:
            raise NotImplementedError("cross_modulator is never enabled by get_arch (SURVEY.md section 2 row 10)")
        if token_mlp != 'leff':
            raise NotImplementedError("only token_mlp='leff' is on the hot path (utils/model_utils.py:65-78)")
        if norm_layer is not nn.LayerNorm or drop:
            raise NotImplementedError("LayerNorm / drop=0 only")
        self.dim = dim
        self.input_resolution = tuple(input_resolution)
        self.num_heads = num_heads
        self.win_size = win_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        self.token_mlp = token_mlp
        if min(self.input_resolution) <= self.win_size:   # model.py:863-866
            self.shift_size = 0
            self.win_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.win_size, "shift_size must in 0-win_size"
        if self.win_size != 8 or win_size != 8:
            raise NotImplementedError("the HIP path is built for win_size 8 (constructor resolution must be >= 8)")
        self.modulator = nn.Embedding(win_size * win_size, dim) if modulator else None
        self.cross_modulator = None
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, win_size=(self.win_size, self.win_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, token_projection=token_projection)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = LeFF(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self._packed = None

    def extra_repr(self) -> str:
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"win_size={self.win_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")

    def _pack(self, dtype):
        sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
        key = (dtype, tuple((v.data_ptr(), v._version) for v in sd.values()))
        if self._packed is None or self._packed[0] != key:
            bp, keep = packing.pack_block(sd, "", self.num_heads, self.shift_size, dtype)
            self._packed = (key, bp, keep)
        return self._packed[1]

    def user_attn_mask(self, mask: Tensor, H: int, W: int) -> Tensor:
        """Dense additive mask (B*nW,64,64) from the user ``mask`` argument, model.py:914-921
        (index glue on the GPU: nearest interpolate, partition, outer product, +-100 fill)."""
        m = torch.nn.functional.interpolate(mask.float(), size=(H, W)).permute(0, 2, 3, 1).contiguous()
        mw = ops.window_partition(m, 8, 0).reshape(-1, 64)
        am = mw.unsqueeze(2) * mw.unsqueeze(1)
        return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am)).contiguous()

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, compute_dtype=torch.float32) -> Tensor:
        """(B, L, C) -> (B, L, C).  With grad mode on and something to differentiate (train() mode, an input or a parameter that
        requires grad) the block is an autograd node like the reference's (model.py:908-989): op-by-op forward that keeps its
        intermediates + the op-level backward (uformer_amd.train.LeWinBlockFunction), timm's DropPath per sample in train() mode.
        Otherwise: the fused inference kernels (DropPath is the identity)."""
        if torch.is_grad_enabled() and mask is None and (self.training or x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import train
            if not x.is_cuda:
                raise UformerHipError("LeWinTransformerBlock runs on the GPU only; there is no CPU path")
            named = [(n, t) for n, t in self.state_dict(keep_vars=True).items()]
            drop = None
            if self.training and self.drop_path_rate > 0.0:
                drop = getattr(self, "_drop_scales_override", None)
                if drop is None:
                    drop = train.sample_drop_scales([self.drop_path_rate], x.shape[0], x.device)
            return train.LeWinBlockFunction.apply(x, [n for n, _ in named], self.num_heads, self.shift_size, compute_dtype, drop, *[t for _, t in named])
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("the mask argument is not supported by the block's autograd path (no reference script passes it)")
        B, L, Cc = x.shape
        H = W = int(math.sqrt(L))
        if not x.is_cuda:
            raise UformerHipError("LeWinTransformerBlock runs on the GPU only; there is no CPU path")
        dt = ops.uf_dtype(compute_dtype)
        y = x.detach().float().contiguous().clone()
        um = self.user_attn_mask(mask, H, W) if mask is not None else None
        with torch.cuda.device(x.device):
            lib = _lib.load()
            nbytes = lib.uf_block_workspace_bytes(B * L, Cc, dt)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            bp = self._pack(compute_dtype)
            _lib.check(lib.uf_lewin_block_fwd(bp, y.data_ptr(), Cc, B, H, W, Cc, None if um is None else um.data_ptr(),
                                              0 if um is None else um.shape[0], dt, ws.data_ptr(), nbytes,
                                              torch.cuda.current_stream().cuda_stream), "uf_lewin_block_fwd")
        return y.to(x.dtype)


class BasicUformerLayer(nn.Module):
    """model.py:1013-1066."""

    def __init__(self, dim, output_dim, input_resolution, depth, num_heads, win_size, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, use_checkpoint=False,
                 token_projection='linear', token_mlp='ffn', shift_flag=True, modulator=False, cross_modulator=False):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            LeWinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, win_size=win_size,
                                  shift_size=0 if (i % 2 == 0 or not shift_flag) else win_size // 2,
                                  mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                  drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                  norm_layer=norm_layer, token_projection=token_projection, token_mlp=token_mlp,
                                  modulator=modulator, cross_modulator=cross_modulator)
            for i in range(depth)])

    def forward(self, x, mask=None, compute_dtype=torch.float32):
        for blk in self.blocks:
            x = blk(x, None if self.use_checkpoint else mask, compute_dtype)  # checkpoint drops mask, model.py:1057
        return x


class Uformer(nn.Module):
    """model.py:1069-1328.  Same kwargs; extra keyword ``compute_dtype`` (default bf16)."""

    def __init__(self, img_size=256, in_chans=3, dd_in=3, embed_dim=32, depths=[2, 2, 2, 2, 2, 2, 2, 2, 2],
                 num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True,
                 use_checkpoint=False, token_projection='linear', token_mlp='leff', dowsample=Downsample, upsample=Upsample,
                 shift_flag=True, modulator=False, cross_modulator=False, compute_dtype=torch.bfloat16, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("drop_rate / attn_drop_rate are 0 in every reference arch")
        if dowsample is not Downsample or upsample is not Upsample:
            raise NotImplementedError("custom sampler classes are not supported")
        self.num_enc_layers = len(depths) // 2
        self.num_dec_layers = len(depths) // 2
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.mlp_ratio = mlp_ratio
        self.token_projection = token_projection
        self.mlp = token_mlp
        self.win_size = win_size
        self.reso = img_size
        self.dd_in = dd_in
        self.in_chans = in_chans
        self.use_checkpoint = use_checkpoint
        self.compute_dtype = compute_dtype
        self.cfg = UformerConfig(img_size=img_size, in_chans=in_chans, dd_in=dd_in, embed_dim=embed_dim,
                                 depths=tuple(depths), num_heads=tuple(num_heads), win_size=win_size, mlp_ratio=mlp_ratio,
                                 modulator=modulator, shift_flag=shift_flag)
        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:self.num_enc_layers]))]
        conv_dpr = [drop_path_rate] * depths[4]
        dec_dpr = enc_dpr[::-1]

        def layer(s, dim, res, dpr, mod):
            return BasicUformerLayer(dim=dim, output_dim=dim, input_resolution=(res, res), depth=depths[s],
                                     num_heads=num_heads[s], win_size=win_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr,
                                     norm_layer=norm_layer, use_checkpoint=use_checkpoint, token_projection=token_projection,
                                     token_mlp=token_mlp, shift_flag=shift_flag, modulator=mod, cross_modulator=False)

        if cross_modulator:
            raise NotImplementedError("cross_modulator is never enabled by get_arch")
        e = embed_dim
        self.input_proj = InputProj(in_channel=dd_in, out_channel=e, kernel_size=3, stride=1, act_layer=nn.LeakyReLU)
        self.output_proj = OutputProj(in_channel=2 * e, out_channel=in_chans, kernel_size=3, stride=1)
        self.encoderlayer_0 = layer(0, e, img_size, enc_dpr[sum(depths[:0]):sum(depths[:1])], False)
        self.dowsample_0 = dowsample(e, e * 2)
        self.encoderlayer_1 = layer(1, e * 2, img_size // 2, enc_dpr[sum(depths[:1]):sum(depths[:2])], False)
        self.dowsample_1 = dowsample(e * 2, e * 4)
        self.encoderlayer_2 = layer(2, e * 4, img_size // 4, enc_dpr[sum(depths[:2]):sum(depths[:3])], False)
        self.dowsample_2 = dowsample(e * 4, e * 8)
        self.encoderlayer_3 = layer(3, e * 8, img_size // 8, enc_dpr[sum(depths[:3]):sum(depths[:4])], False)
        self.dowsample_3 = dowsample(e * 8, e * 16)
        self.conv = layer(4, e * 16, img_size // 16, conv_dpr, False)
        self.upsample_0 = upsample(e * 16, e * 8)
        self.decoderlayer_0 = layer(5, e * 16, img_size // 8, dec_dpr[:depths[5]], modulator)
        self.upsample_1 = upsample(e * 16, e * 4)
        self.decoderlayer_1 = layer(6, e * 8, img_size // 4, dec_dpr[sum(depths[5:6]):sum(depths[5:7])], modulator)
        self.upsample_2 = upsample(e * 8, e * 2)
        self.decoderlayer_2 = layer(7, e * 4, img_size // 2, dec_dpr[sum(depths[5:7]):sum(depths[5:8])], modulator)
        self.upsample_3 = upsample(e * 4, e)
        self.decoderlayer_3 = layer(8, e * 2, img_size, dec_dpr[sum(depths[5:8]):sum(depths[5:9])], modulator)
        self.apply(self._init_weights)
        self._packed = None
        self._packed_key = None
        self._ws = None

    def _init_weights(self, m):   # model.py:1249-1256
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def extra_repr(self) -> str:
        return (f"embed_dim={self.embed_dim}, token_projection={self.token_projection}, token_mlp={self.mlp},"
                f"win_size={self.win_size}, compute_dtype={self.compute_dtype}")

    # ---- packed weights ------------------------------------------------------------------
    def _apply(self, fn, *a, **k):   # .to() / .cuda() / .half(): parameters are replaced -- packed weights, workspaces and the cached parameter list go
        self._packed = None
        self._ws = None
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts reference checkpoints, including ``{'state_dict': ...}`` payloads and the
        ``module.`` prefix left by nn.DataParallel (utils/model_utils.py:23-33)."""
        if "state_dict" in state_dict and not any(k.startswith("input_proj") for k in state_dict):
            state_dict = state_dict["state_dict"]
        if all(k.startswith("module.") for k in state_dict):
            state_dict = {k[7:]: v for k, v in state_dict.items()}
        self._packed = None
        self.__dict__.pop("_plist", None)
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def repack(self):
        self._packed = None
        self.__dict__.pop("_plist", None)

    def _get_packed(self, device):
        self._check_not_replica()
        # (storage, version) of every parameter: catches in-place updates (optimizer steps bump _version) AND writes through
        # ``p.data`` / ``p.data = ...`` that swap the storage without bumping it (EMA swaps, older optimizers)
        # The parameter LIST is cached (walking the module tree costs 1.1 ms per call for Uformer-B's 719 parameters, the key itself
        # 0.1 ms): Parameter objects survive .to() / load_state_dict() / optimizer steps; replacing one (m.w = nn.Parameter(..)) needs
        # repack(), which also drops this list -- as do _apply() and load_state_dict().
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        key = (self.compute_dtype, str(device), tuple(map(torch.Tensor.data_ptr, plist)), tuple(p._version for p in plist))
        if self._packed is None or self._packed_key != key:
            sd = self.state_dict(keep_vars=True)
            self._packed = packing.PackedModel(self.cfg, sd, self.compute_dtype)
            self._packed_key = key
        return self._packed

    def _check_not_replica(self):
        """nn.DataParallel replicas carry no parameters of their own (``_parameters`` is emptied by ``replicate``) and would
        share the packed weights / workspace of device 0.  The reference wraps the model in nn.DataParallel
        (train/train_denoise.py:83); here multi-GPU is one process per GPU (uformer_amd.dist) -- say so instead of failing
        with a KeyError deep inside the packing code."""
        if getattr(self, "_is_replica", False):
            raise UformerHipError("uformer_amd.Uformer cannot run as an nn.DataParallel replica: launch one process per GPU "
                                  "(torch.distributed.run, uformer_amd.dist.shard_batch / GradientAllReduce) instead; "
                                  "nn.DataParallel over a single device is fine")

    def _workspace(self, need: int, device) -> Tensor:
        """One workspace per (device, caller stream): two host threads driving the same module on their own streams must
        not scribble over each other's activations."""
        if self._ws is None:
            from collections import OrderedDict
            self._ws = OrderedDict()
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=device)
        self._ws.move_to_end(key)
        while len(self._ws) > self.MAX_WORKSPACES:       # least recently used first: a caller cycling through PyTorch's stream pool
            self._ws.popitem(last=False)                 # (32 per priority) must not pin a multi-GB workspace per stream; the block goes
                                                         # back to the caching allocator's pool of the stream it was allocated (and used) on
        return ws

    MAX_WORKSPACES = 8      # (round 6: infer.PipelinedForward drives the module from a ring of streams)

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        if not x.is_cuda:
            raise UformerHipError("uformer_amd.Uformer runs on an MI355X only; there is no CPU fallback "
                                  "(the CPU oracle lives in oracle/ and is test infrastructure)")
        if x.dim() != 4 or x.shape[1] != self.dd_in:
            raise UformerHipError(f"expected (B,{self.dd_in},H,W) input, got {tuple(x.shape)}")
        # autograd is wanted whenever grad mode is on and something upstream requires it -- in train() AND in eval() mode
        # (fine-tuning with frozen statistics, input gradients): the fused inference kernels keep no activations and would
        # return a tensor without grad_fn, i.e. silently drop the data term's gradients
        if torch.is_grad_enabled() and (self.training or x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if mask is None:
                return self._forward_train(x)
            # the autograd path does not take the mask argument (no reference script passes it).  Gradients explicitly asked for
            # (train() mode, or an input that requires grad): say so.  eval() with grad mode merely left on (``model.eval(); model(x,
            # mask)`` without torch.no_grad(), the reference's test scripts' habit): run the inference kernels, warn once.
            if self.training or x.requires_grad:
                raise NotImplementedError("the mask argument is not supported by the autograd path (no reference script passes it): "
                                          "wrap the call in torch.no_grad() for inference")
            if not getattr(self, "_warned_eval_mask", False):
                import warnings
                warnings.warn("uformer_amd.Uformer: eval() forward with a mask while grad mode is on runs the inference kernels; the result has no "
                              "grad_fn (wrap the call in torch.no_grad() to silence this)", stacklevel=2)
                self._warned_eval_mask = True
        if mask is not None:
            return self._forward_blockwise(x, mask)
        B, _, H, W = x.shape
        xin = x.detach().float().contiguous()
        dt = ops.uf_dtype(self.compute_dtype)
        with torch.cuda.device(x.device):
            lib = _lib.load()
            pk = self._get_packed(x.device)
            need = lib.uf_uformer_workspace_bytes(pk.desc, B, H, W, dt)
            if need == 0:
                raise UformerHipError("uf_uformer_workspace_bytes: " + _lib.last_error())
            ws = self._workspace(need, x.device)
            out = torch.empty((B, self.in_chans, H, W), dtype=torch.float32, device=x.device)
            _lib.check(lib.uf_uformer_fwd(pk.desc, xin.data_ptr(), out.data_ptr(), B, H, W, dt, ws.data_ptr(),
                                          ws.numel(), torch.cuda.current_stream().cuda_stream), "uf_uformer_fwd")
        return out.to(x.dtype)

    def drop_path_rates(self):
        """Per-block stochastic-depth rates in execution order (model.py:1093-1095)."""
        from .spec import STAGES
        return [float(blk.drop_path_rate) for name in STAGES for blk in getattr(self, name).blocks]

    def _forward_train(self, x: Tensor) -> Tensor:
        """train() with grad enabled: the op-by-op tape of uformer_amd/train.py behind torch.autograd (the fused inference
        kernels keep no activations).  DropPath masks are drawn here, per block and branch, as timm's DropPath does
        (train/train_denoise.py:181-184 then calls backward on the loss)."""
        from . import train
        self._check_not_replica()
        sd = self.state_dict(keep_vars=True)
        names, params = list(sd.keys()), list(sd.values())
        sink = getattr(self, "grad_sink", None)
        # gradients go straight into the all-reduce buckets as the reverse sweep finishes each stage (sink); use_checkpoint=True
        # (model.py:1056-1057: torch.utils.checkpoint around every block) = the recompute form: a block keeps only its input
        names = train.NamesWithSink(names)
        names.sink = sink
        # (2-byte operand types only: compute_dtype=float32 always keeps its intermediates -- UformerTape's rule; the f32 fused forward + op-level
        #  recomputation pair is a combination no fixture covers)
        names.recompute = True if (self.use_checkpoint and self.compute_dtype in (torch.bfloat16, torch.float16)) else None
        rates = self.drop_path_rates() if self.training else []      # eval(): DropPath is the identity (timm)
        drop = getattr(self, "_drop_scales_override", None) if self.training else None
        if drop is None and any(r > 0 for r in rates):
            drop = train.sample_drop_scales(rates, x.shape[0], x.device)
        return train.UformerFunction.apply(x, self.cfg, self.compute_dtype, drop, names, *params)

    def _forward_blockwise(self, x: Tensor, mask: Optional[Tensor]) -> Tensor:
        """Module-by-module path (used when the rarely-used ``mask`` argument is given):
        the same wiring as model.py:1269-1305, every step through the C ABI."""
        cd = self.compute_dtype
        y = self.input_proj(x.detach().float())
        conv0 = self.encoderlayer_0(y, mask, cd)
        conv1 = self.encoderlayer_1(self.dowsample_0(conv0, cd), mask, cd)
        conv2 = self.encoderlayer_2(self.dowsample_1(conv1, cd), mask, cd)
        conv3 = self.encoderlayer_3(self.dowsample_2(conv2, cd), mask, cd)
        conv4 = self.conv(self.dowsample_3(conv3, cd), mask, cd)
        d0 = self.decoderlayer_0(torch.cat([self.upsample_0(conv4, cd), conv3], -1), mask, cd)
        d1 = self.decoderlayer_1(torch.cat([self.upsample_1(d0, cd), conv2], -1), mask, cd)
        d2 = self.decoderlayer_2(torch.cat([self.upsample_2(d1, cd), conv1], -1), mask, cd)
        d3 = self.decoderlayer_3(torch.cat([self.upsample_3(d2, cd), conv0], -1), mask, cd)
        out = self.output_proj(d3, x.detach().float() if self.dd_in == 3 else None)
        return out.to(x.dtype)

    def flops(self):
        """Exact multiply-accumulates of one forward at the constructor resolution."""
        total = 0
        r = self.reso
        total += self.input_proj.flops(r, r) + self.output_proj.flops(r, r)
        dims = self.cfg.stage_dims()
        div = self.cfg.stage_res_div()
        for s in range(9):
            L = (r // div[s]) ** 2
            total += self.cfg.depths[s] * L * dims[s] * (12 * dims[s] + 2 * 64 + 36)
        for s in range(4):
            L = (r // div[s]) ** 2
            total += (L // 4) * dims[s] * 2 * dims[s] * 16
        for k, (cin, cout) in enumerate(self.cfg.upsample_io()):
            L = (r // div[4 + k]) ** 2
            total += 4 * L * cin * cout
        return total


def get_arch(arch: str, train_ps: int = 128, dd_in: int = 3, embed_dim: int = 32, compute_dtype=torch.bfloat16) -> Uformer:
    """utils/model_utils.py:56-81 ``get_arch(opt)`` with the option fields as arguments."""
    common = dict(win_size=8, token_projection='linear', token_mlp='leff', modulator=True, compute_dtype=compute_dtype)
    if arch == 'Uformer':
        return Uformer(img_size=train_ps, embed_dim=embed_dim, **common)
    if arch == 'Uformer_T':
        return Uformer(img_size=train_ps, embed_dim=16, **common)
    if arch == 'Uformer_S':
        return Uformer(img_size=train_ps, embed_dim=32, **common)
    if arch == 'Uformer_S_noshift':
        return Uformer(img_size=train_ps, embed_dim=32, shift_flag=False, **common)
    if arch == 'Uformer_B':
        return Uformer(img_size=train_ps, embed_dim=32, depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], dd_in=dd_in, **common)
    raise Exception("Arch error!")
