"""CPU oracle for the Uformer LeWin-block hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain, functional, fp32 CPU restatement of the arithmetic of the
reference's ``model.py`` for the hot path named in BASELINE.json:
window partition / reverse, cyclic shift + SW-MSA mask, LayerNorm, the modulator
add, LinearProjection + WindowAttention with relative-position bias, LeFF,
Downsample / Upsample, Input/OutputProj and the U-shaped wiring of ``Uformer``.

It exists to CHECK the HIP product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``uformer_amd/`` does, and the product path raises if
its HIP library is missing rather than falling back to anything here.

Parity pin: the reference ships no golden vectors for this path (SURVEY.md §4,
§8c), so the oracle is pinned against the reference ITSELF: the script
``tests/golden/make_golden.py`` imports ``/root/reference/model.py`` (unmodified,
with a 3-symbol ``timm`` shim), runs it on seeded weights/inputs and commits the
outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against those fixtures (bit-exact for the index ops, <=2e-5 abs
for floating point -- the only difference is summation order inside ATen).

Every function cites the reference ``file:line`` it restates.  All tensors are
``torch.float32`` on CPU (index helpers are numpy / int64).  The weights are
taken from a reference-layout ``state_dict``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

WIN = 8  # reference fixes win_size=8 in every shipped arch (utils/model_utils.py:65-78)


# ----------------------------------------------------------------------------------------
# index-only ops (bit-exact)
# ----------------------------------------------------------------------------------------
def window_partition(x: Tensor, win: int = WIN) -> Tensor:
    """(B,H,W,C) -> (B*nW, win, win, C).  model.py:704-715 (non-dilated branch :713-714).

    Closed form: out[b*nW + (h//win)*(W//win) + w//win, h%win, w%win, c] = x[b,h,w,c].
    """
    B, H, W, C = x.shape
    x = x.reshape(B, H // win, win, W // win, win, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().reshape(-1, win, win, C)


def window_reverse(windows: Tensor, win: int, H: int, W: int) -> Tensor:
    """(B*nW, win, win, C) -> (B,H,W,C).  model.py:717-726 (:720,725)."""
    nW = (H // win) * (W // win)
    B = windows.shape[0] // nW
    x = windows.reshape(B, H // win, W // win, win, win, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().reshape(B, H, W, -1)


def window_partition_index(B: int, H: int, W: int, win: int = WIN, shift: int = 0) -> np.ndarray:
    """Flat source-token index for every window-order token (numpy int64).

    idx[m] = flat index into (B,H,W) of the token that lands at window-order row m after
    ``torch.roll(x, (-shift,-shift), (1,2))`` (model.py:957) followed by ``window_partition``.
    Used by the tests to check the HIP index math without any floating point.
    """
    b, wr, wc, y, x = np.meshgrid(np.arange(B), np.arange(H // win), np.arange(W // win),
                                  np.arange(win), np.arange(win), indexing="ij")
    h = (wr * win + y + shift) % H   # roll by -shift: shifted[h] = x[(h+shift) % H]
    w = (wc * win + x + shift) % W
    return (b * H * W + h * W + w).reshape(-1).astype(np.int64)


def relative_position_index(win: int = WIN) -> np.ndarray:
    """(win*win, win*win) int64.  model.py:467-477.

    idx[i,j] = (yi-yj+win-1)*(2*win-1) + (xi-xj+win-1), tokens row-major in the window.
    """
    ys, xs = np.meshgrid(np.arange(win), np.arange(win), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + win - 1
    dx = xs[:, None] - xs[None, :] + win - 1
    return (dy * (2 * win - 1) + dx).astype(np.int64)


def shift_attn_mask(H: int, W: int, win: int = WIN, shift: int = WIN // 2) -> Tensor:
    """SW-MSA additive mask (nW, win*win, win*win), values in {0,-100}.  model.py:924-942."""
    img = torch.zeros((1, H, W, 1), dtype=torch.float32)
    slices = (slice(0, -win), slice(-win, -shift), slice(-shift, None))
    cnt = 0
    for hs in slices:
        for ws in slices:
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = window_partition(img, win).reshape(-1, win * win)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def input_attn_mask(mask: Tensor, H: int, W: int, win: int = WIN) -> Tensor:
    """User ``mask`` path of the block.  model.py:914-921 (both-tokens-nonzero => -100)."""
    m = F.interpolate(mask, size=(H, W)).permute(0, 2, 3, 1)
    mw = window_partition(m, win).reshape(-1, win * win)
    am = mw.unsqueeze(2) * mw.unsqueeze(1)
    return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))


# ----------------------------------------------------------------------------------------
# floating-point pieces
# ----------------------------------------------------------------------------------------
def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() exact erf form (model.py:657-660 use the default)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(C) over the last dim, biased variance.  model.py:881,888."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def relative_position_bias(table: Tensor, index: Tensor) -> Tensor:
    """(heads, N, N) dense bias from the (225, heads) table.  model.py:500-502."""
    N = index.shape[0]
    return table[index.reshape(-1)].reshape(N, N, -1).permute(2, 0, 1).contiguous()


def window_attention(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int,
                     mask: Optional[Tensor] = None) -> Tensor:
    """WindowAttention.forward for self-attention.  model.py:494-522 + LinearProjection :431-442.

    x: (B_, N, C).  mask: (nW, N, N) additive or None.
    """
    B_, N, C = x.shape
    hd = C // heads
    q = x @ p[prefix + "qkv.to_q.weight"].t() + p[prefix + "qkv.to_q.bias"]
    kv = x @ p[prefix + "qkv.to_kv.weight"].t() + p[prefix + "qkv.to_kv.bias"]
    q = q.reshape(B_, N, heads, hd).permute(0, 2, 1, 3)               # :438,440
    k = kv[..., :C].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)     # :439 k = channels [0,C)
    v = kv[..., C:].reshape(B_, N, heads, hd).permute(0, 2, 1, 3)     #      v = channels [C,2C)
    q = q * (hd ** -0.5)                                              # :497
    attn = q @ k.transpose(-2, -1)                                    # :498
    bias = relative_position_bias(p[prefix + "relative_position_bias_table"],
                                  p[prefix + "relative_position_index"])
    attn = attn + bias.unsqueeze(0)                                   # :506
    if mask is not None:                                              # :508-512
        nW = mask.shape[0]
        attn = attn.reshape(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)                                # :513-515
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)                # :519
    return out @ p[prefix + "proj.weight"].t() + p[prefix + "proj.bias"]  # :520


def leff(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """LeFF.forward.  model.py:666-685 (modules :657-661).  x: (B, L, C), L square."""
    B, L, C = x.shape
    hh = int(math.sqrt(L))
    h = gelu_erf(x @ p[prefix + "linear1.0.weight"].t() + p[prefix + "linear1.0.bias"])
    hid = h.shape[-1]
    h = h.reshape(B, hh, hh, hid).permute(0, 3, 1, 2)                 # :674
    h = F.conv2d(h, p[prefix + "dwconv.0.weight"], p[prefix + "dwconv.0.bias"],
                 stride=1, padding=1, groups=hid)                     # :659
    h = gelu_erf(h)
    h = h.permute(0, 2, 3, 1).reshape(B, L, hid)                      # :680
    return h @ p[prefix + "linear2.0.weight"].t() + p[prefix + "linear2.0.bias"]


def lewin_block(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, shift: int,
                win: int = WIN, mask: Optional[Tensor] = None, drop: Optional[Tensor] = None) -> Tensor:
    """LeWinTransformerBlock.forward.  model.py:908-989.  ``drop`` = None: eval mode (DropPath = identity); else a (2, B)
    tensor of per-sample scales bernoulli(keep)/keep for the attention and the LeFF branch (timm DropPath, :986-987)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    attn_mask = input_attn_mask(mask, H, W, win) if mask is not None else None   # :914-921
    if shift > 0:                                                                # :924-942
        sm = shift_attn_mask(H, W, win, shift)
        attn_mask = attn_mask + sm if attn_mask is not None else sm
    shortcut = x
    y = layer_norm(x, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])      # :952
    y = y.reshape(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))                  # :957
    yw = window_partition(y, win).reshape(-1, win * win, C)                      # :962-963
    if (prefix + "modulator.weight") in p:                                       # :966-969
        yw = yw + p[prefix + "modulator.weight"]
    aw = window_attention(yw, p, prefix + "attn.", heads, attn_mask)             # :972
    y = window_reverse(aw.reshape(-1, win, win, C), win, H, W)                   # :975-976
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))                    # :980
    y = y.reshape(B, L, C)
    if drop is not None:
        y = y * drop[0].reshape(B, 1, 1)
    x = shortcut + y                                                             # :986
    z = layer_norm(x, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])
    m = leff(z, p, prefix + "mlp.")
    if drop is not None:
        m = m * drop[1].reshape(B, 1, 1)
    return x + m                                                                 # :987


def downsample(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """Downsample.forward: Conv2d(C,2C,k4,s2,p1) on tokens.  model.py:739-746 (:734)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    y = x.transpose(1, 2).reshape(B, C, H, W)
    y = F.conv2d(y, p[prefix + "conv.0.weight"], p[prefix + "conv.0.bias"], stride=2, padding=1)
    return y.flatten(2).transpose(1, 2).contiguous()


def upsample(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """Upsample.forward: ConvTranspose2d(Cin,Cout,k2,s2) on tokens.  model.py:765-771 (:760)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    y = x.transpose(1, 2).reshape(B, C, H, W)
    y = F.conv_transpose2d(y, p[prefix + "deconv.0.weight"], p[prefix + "deconv.0.bias"], stride=2)
    return y.flatten(2).transpose(1, 2).contiguous()


def input_proj(x: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """InputProj: conv3x3 + LeakyReLU(0.01) -> tokens.  model.py:795-800 (:785-786)."""
    y = F.conv2d(x, p["input_proj.proj.0.weight"], p["input_proj.proj.0.bias"], stride=1, padding=1)
    y = torch.where(y >= 0, y, 0.01 * y)
    return y.flatten(2).transpose(1, 2).contiguous()


def output_proj(x: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """OutputProj: tokens -> conv3x3.  model.py:828-836 (:817)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    y = x.transpose(1, 2).reshape(B, C, H, W)
    return F.conv2d(y, p["output_proj.proj.0.weight"], p["output_proj.proj.0.bias"], stride=1, padding=1)


# ----------------------------------------------------------------------------------------
# model wiring
# ----------------------------------------------------------------------------------------
STAGES = ("encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
          "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3")


def block_shifts(img_size: int, depths: Sequence[int], win: int = WIN) -> List[List[int]]:
    """Per-stage, per-block shift sizes exactly as the constructors decide them.

    ``shift = 0 if i%2==0 else win//2`` (model.py:1030), then clamped to 0 when the
    CONSTRUCTOR resolution ``<= win`` (model.py:863-866).
    """
    res = [img_size, img_size // 2, img_size // 4, img_size // 8, img_size // 16,
           img_size // 8, img_size // 4, img_size // 2, img_size]
    out = []
    for s, d in enumerate(depths):
        row = []
        for i in range(d):
            sh = 0 if i % 2 == 0 else win // 2
            if res[s] <= win:
                if res[s] < win:
                    raise NotImplementedError("oracle: constructor resolution < win_size (win clamp) not restated")
                sh = 0
            row.append(sh)
        out.append(row)
    return out


def uformer_forward(x: Tensor, p: Dict[str, Tensor], *, img_size: int, embed_dim: int,
                    depths: Sequence[int], num_heads: Sequence[int], win: int = WIN,
                    dd_in: int = 3, mask: Optional[Tensor] = None, drop_scales: Optional[Tensor] = None) -> Tensor:
    """Uformer.forward.  model.py:1269-1305.  ``p`` = reference state_dict.  ``drop_scales`` = None: eval mode; else the
    (2 * n_blocks, B) DropPath scales in execution order (two rows per block: attention branch, LeFF branch)."""
    shifts = block_shifts(img_size, depths, win)
    first = [sum(depths[:s]) for s in range(9)]

    def stage(y: Tensor, s: int) -> Tensor:
        for i in range(depths[s]):                                   # model.py:1054-1060
            bi = first[s] + i
            dr = drop_scales[2 * bi:2 * bi + 2] if drop_scales is not None else None
            y = lewin_block(y, p, f"{STAGES[s]}.blocks.{i}.", num_heads[s], shifts[s][i], win, mask, dr)
        return y

    y = input_proj(x, p)                                             # :1271
    conv0 = stage(y, 0)
    pool0 = downsample(conv0, p, "dowsample_0.")
    conv1 = stage(pool0, 1)
    pool1 = downsample(conv1, p, "dowsample_1.")
    conv2 = stage(pool1, 2)
    pool2 = downsample(conv2, p, "dowsample_2.")
    conv3 = stage(pool2, 3)
    pool3 = downsample(conv3, p, "dowsample_3.")
    conv4 = stage(pool3, 4)                                          # :1284
    up0 = upsample(conv4, p, "upsample_0.")
    d0 = stage(torch.cat([up0, conv3], -1), 5)                       # :1288 (upsampled FIRST)
    up1 = upsample(d0, p, "upsample_1.")
    d1 = stage(torch.cat([up1, conv2], -1), 6)
    up2 = upsample(d1, p, "upsample_2.")
    d2 = stage(torch.cat([up2, conv1], -1), 7)
    up3 = upsample(d2, p, "upsample_3.")
    d3 = stage(torch.cat([up3, conv0], -1), 8)
    y = output_proj(d3, p)                                           # :1304
    return x + y if dd_in == 3 else y                                # :1305


def expand2square(img: Tensor, factor: float = 128.0):
    """Arbitrary-resolution wrapper of the eval scripts.  test/test_sidd.py:79-92."""
    _, _, h, w = img.shape
    X = int(math.ceil(max(h, w) / float(factor)) * factor)
    out = torch.zeros(1, 3, X, X, dtype=img.dtype)
    msk = torch.zeros(1, 1, X, X, dtype=img.dtype)
    out[:, :, (X - h) // 2:(X - h) // 2 + h, (X - w) // 2:(X - w) // 2 + w] = img
    msk[:, :, (X - h) // 2:(X - h) // 2 + h, (X - w) // 2:(X - w) // 2 + w].fill_(1)
    return out, msk


def charbonnier_loss(x: Tensor, y: Tensor, eps: float = 1e-3) -> Tensor:
    """CharbonnierLoss.forward: mean(sqrt((x-y)^2 + eps^2)).  losses.py:41-52 (training criterion,
    train/train_denoise.py:164,181)."""
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps * eps))


def psnr(a: Tensor, b: Tensor) -> float:
    """myPSNR: 20*log10(1/rmse) on clamped tensors.  utils/image_utils.py:40-44."""
    d = torch.clamp(a, 0, 1) - torch.clamp(b, 0, 1)
    rmse = float((d ** 2).mean().sqrt())
    return float("inf") if rmse == 0 else 20.0 * math.log10(1.0 / rmse)


# ---- the steps either side of the path: metrics, augmentation -- restated for the tests ------------------------------
def batch_psnr(img1: Tensor, img2: Tensor, average: bool = True) -> float:
    """batch_PSNR: per-image myPSNR summed / averaged.  utils/image_utils.py:46-51."""
    ps = [psnr(a, b) for a, b in zip(img1, img2)]
    return sum(ps) / len(ps) if average else sum(ps)


def ssim(img1: Tensor, img2: Tensor) -> float:
    """calculate_ssim on float (C,H,W) images in [0,1]: x255, round, uint8; per channel the 11x11 Gaussian (sigma 1.5) statistics
    on the valid region (cv2.filter2D(...)[5:-5, 5:-5] == valid correlation), mean over the map, mean over channels.
    utils/caculate_psnr_ssim.py:35-81 (crop_border 0, test_y_channel False)."""
    import numpy as np
    from scipy.signal import correlate2d
    # (img * 255.0).round().astype(np.uint8) exactly as the reference writes it (:59-62): no clamp -- out-of-range values wrap modulo 256, as the
    # float -> uint8 conversion does (spelled through int64 so that the wrap does not depend on the platform's float -> uint8 behaviour)
    a = ((img1.numpy() * 255.0).round().astype(np.int64) & 255).astype(np.float64)        # float32 product and round, as the reference's
    b = ((img2.numpy() * 255.0).round().astype(np.int64) & 255).astype(np.float64)
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()                                     # cv2.getGaussianKernel(11, 1.5)
    window = np.outer(k, k)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(a.shape[0]):
        f = lambda z: correlate2d(z, window, mode="valid")   # noqa: E731
        mu1, mu2 = f(a[c]), f(b[c])
        s1, s2, s12 = f(a[c] ** 2) - mu1 ** 2, f(b[c] ** 2) - mu2 ** 2, f(a[c] * b[c]) - mu1 * mu2
        m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
        vals.append(m.mean())
    return float(np.mean(vals))


def augment(x: Tensor, k: int) -> Tensor:
    """Augment_RGB_torch.transform<k> (utils/dataset_utils.py:8-33): rot90 by k & 3 in dims [-1,-2], then flip(-2) for k >= 4."""
    y = torch.rot90(x, k=k & 3, dims=[-1, -2])
    return y.flip(-2) if k >= 4 else y


def crop_augment(frame_chw: Tensor, r: int, c: int, ps: int, k: int) -> Tensor:
    """DataLoaderTrain.__getitem__ after loading: crop [r:r+ps, c:c+ps] then the transform.  dataset/dataset_denoise.py:54-70."""
    return augment(frame_chw[:, r:r + ps, c:c + ps], k)


def mixup(x: Tensor, lam: Tensor, perm: Tensor) -> Tensor:
    """MixUp_AUG.aug for one tensor: lam x + (1 - lam) x[perm].  utils/dataset_utils.py:44-53."""
    lam = lam.view(-1, 1, 1, 1)
    return lam * x + (1 - lam) * x[perm]
