"""GPU parity tests, op by op, through the C ABI (uformer_amd.ops -> libuformer_hip.so).

Checker = oracle/uformer_oracle.py (pinned to the reference by tests/test_oracle_golden.py)
and the committed golden fixtures produced by the reference itself.

Tolerances (stated here, used below):
  * index ops (window partition / reverse / roll, shift mask): BIT-EXACT.
  * f32 mode (exact-f32 MFMA): 1e-3 abs is the north-star gate; per-op we demand 2e-4.
  * bf16 mode (bf16 operands, f32 accumulate): relative to the tensor's max magnitude,
    <= 2.5e-2 per op (bf16 has 8 mantissa bits; two roundings of O(1) values).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import uformer_oracle as O

pytestmark = pytest.mark.gpu

F32_TOL = 2e-4
BF16_REL = 2.5e-2
F16_REL = BF16_REL / 8      # three more mantissa bits than bf16
MODES = [torch.float32, torch.bfloat16, torch.float16]
TAG = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}
REPORT = {}


def t(a):
    return torch.from_numpy(np.asarray(a))


def params(g, prefix):
    return {k[len(prefix):]: t(v) for k, v in g.items() if k.startswith(prefix)}


def cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def check(name, got, ref, dtype):
    got = got.detach().float().cpu()
    ref = ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = {torch.float32: F32_TOL, torch.bfloat16: BF16_REL, torch.float16: F16_REL}[dtype] * max(1.0, scale)
    REPORT[f"{name}[{TAG[dtype]}]"] = {"max_abs_err": err, "ref_max": scale, "tol": tol}
    assert torch.isfinite(got).all(), name
    assert err <= tol, f"{name}: max abs err {err:.3e} > {tol:.3e} (ref max {scale:.3f})"


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_ops.json", "w") as f:
        json.dump(REPORT, f, indent=1)


@pytest.fixture(scope="module")
def ops():
    from uformer_amd import ops as _ops
    return _ops


# ---------------------------------------------------------------------------------------
# bit-exact index ops
# ---------------------------------------------------------------------------------------
def test_window_partition_reverse_golden(ops, golden):
    g = golden("index_ops")
    x = t(g["x"]).cuda()                       # int32 (2,16,24,3): 12-byte rows -> element path
    wp = ops.window_partition(x, 8, 0)
    assert torch.equal(wp.cpu(), t(g["partition"]))
    assert torch.equal(ops.window_reverse(wp, 8, 16, 24, 0).cpu(), t(g["x"]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.int32, torch.int16])
@pytest.mark.parametrize("shape,shift", [((2, 16, 16, 32), 0), ((2, 16, 16, 32), 4), ((1, 32, 24, 8), 4),
                                          ((3, 8, 8, 16), 0), ((1, 128, 128, 64), 4), ((2, 16, 16, 5), 4)])
def test_window_ops_bit_exact(ops, dtype, shape, shift):
    B, H, W, C = shape
    gen = torch.Generator().manual_seed(5)
    if dtype.is_floating_point:
        x = torch.randn(shape, generator=gen).to(dtype)
    else:
        x = torch.randint(-30000, 30000, shape, generator=gen).to(dtype)
    ref = O.window_partition(torch.roll(x, shifts=(-shift, -shift), dims=(1, 2)), 8)
    got = ops.window_partition(x.cuda(), 8, shift)
    assert torch.equal(got.cpu(), ref)
    # closed-form index check (no tensors from the oracle's permute path)
    idx = O.window_partition_index(B, H, W, 8, shift)
    assert torch.equal(got.cpu().reshape(-1, C), x.reshape(-1, C)[torch.from_numpy(idx)])
    back = ops.window_reverse(got, 8, H, W, shift)
    assert torch.equal(back.cpu(), x)           # partition o reverse = identity (round trip)


def test_shift_mask_bit_exact(ops, golden):
    g = golden("index_ops")
    assert torch.equal(ops.shift_mask(16, 16, 4, "cuda").cpu(), t(g["shift_mask_16"]))
    assert torch.equal(ops.shift_mask(32, 32, 4, "cuda").cpu(), t(g["shift_mask_32"]))
    assert torch.equal(ops.shift_mask(64, 40, 4, "cuda").cpu(), O.shift_attn_mask(64, 40, 8, 4))
    assert torch.equal(ops.shift_mask(8, 8, 4, "cuda").cpu(), O.shift_attn_mask(8, 8, 8, 4))
    assert ops.shift_mask(16, 16, 0, "cuda").abs().max().item() == 0


# ---------------------------------------------------------------------------------------
# floating point ops
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("C", [16, 32, 64, 128, 256, 512])
def test_layernorm(ops, dtype, C):
    gen = torch.Generator().manual_seed(C)
    B, H, W = 2, 16, 16
    x = torch.randn(B * H * W, C, generator=gen) * 2 + 0.5
    gm, bt = 1 + 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)
    mod = torch.randn(64, C, generator=gen)
    ref = O.layer_norm(x, gm, bt)
    got = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), B=B, H=H, W=W, dtype=dtype)
    check(f"layernorm_C{C}", got, ref, dtype)
    # norm1 path: roll(-4) + partition + modulator (model.py:952-969)
    y = torch.roll(ref.reshape(B, H, W, C), shifts=(-4, -4), dims=(1, 2))
    refw = O.window_partition(y, 8).reshape(-1, 64, C) + mod
    got = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), B=B, H=H, W=W, dtype=dtype, windowed=True, shift=4,
                        modulator=mod.cuda())
    check(f"layernorm_windowed_C{C}", got, refw.reshape(-1, C), dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 96, 32), (64, 32, 128), (192, 64, 256), (1000, 2048, 512),
                                   (128, 512, 2048), (64, 48, 16), (320, 1536, 512)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear(ops, dtype, M, N, K, act):
    gen = torch.Generator().manual_seed(M + N + K)
    # asymmetric, non-repeating data so a transposed or permuted tile cannot pass
    a = torch.randn(M, K, generator=gen) + torch.arange(M)[:, None] * 1e-3
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    ad, wd = a.to(dtype), w.to(dtype)
    ref = ad.float() @ wd.float().t() + b
    if act:
        ref = O.gelu_erf(ref)
    got = ops.linear(ad.cuda(), wd.cuda(), b.cuda(), act)
    check(f"linear_{M}x{N}x{K}_act{act}", got, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(16384, 1024, 256), (32768, 512, 1024)])
def test_gemm_training_shapes(ops, dtype, M, N, K):
    """Stage-sized products (thousands of 128 x 128 tiles, several K tiles) through every epilogue the training step uses: plain store,
    GELU, pre-activation + GELU, times GELU', and the residual stores (plain and window_reverse with DropPath scales)."""
    gen = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=gen) + torch.arange(M)[:, None] * 1e-5).to(dtype)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=gen)
    ref = a.float() @ w.float().t() + b
    ac, wc, bc = a.cuda(), w.cuda(), b.cuda()
    check(f"big_linear_{M}x{N}x{K}", ops.linear(ac, wc, bc), ref, dtype)
    check(f"big_linear_gelu_{M}x{N}x{K}", ops.linear(ac, wc, bc, 1), O.gelu_erf(ref), dtype)
    pre, act = ops.linear_pre_gelu(ac, wc, bc)
    check(f"big_pre_{M}x{N}x{K}", pre, ref, dtype)
    assert torch.equal(act, ops.gelu(pre))                                         # GELU of the value as stored
    prev = torch.randn(M, N, generator=gen).to(dtype).cuda()
    got = ops.linear_mul_dgelu(ac, wc, bc, prev)
    check(f"big_mul_dgelu_{M}x{N}x{K}", got, ops.gelu_bwd(prev, ops.linear(ac, wc, bc)).float().cpu(), dtype)   # the two-pass form on the same rounded product
    B, H, W = M // 4096, 64, 64
    x = torch.randn(M, N, generator=gen)
    sc = 1.25 * (torch.arange(B) % 3 != 0).float()
    srow = sc.repeat_interleave(H * W).reshape(M, 1)
    tol = 2e-6
    out = ops.linear_residual(ac, wc, bc, x.cuda(), sc.cuda(), B, H, W).cpu()
    assert ((out - (x + srow * ref)).abs().max() / (x + srow * ref).abs().max()).item() < tol
    outw = ops.linear_residual(ac, wc, bc, x.cuda(), sc.cuda(), B, H, W, windowed=True, shift=4).cpu()
    refw = x + srow * ops.window_reverse(ref.reshape(-1, 8, 8, N).cuda(), 8, H, W, 4).reshape(M, N).cpu()
    assert ((outw - refw).abs().max() / refw.abs().max()).item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(4096, 512, 2048), (16384, 256, 1024), (1000, 192, 512), (2048, 1024, 256), (4096, 64, 128), (320, 1536, 512), (130, 96, 128), (4096, 768, 256), (8192, 1536, 512)])
def test_gemm_dma_bit_identical(ops, dtype, M, N, K, monkeypatch):
    """The LDS-DMA staging path of the GEMM (uf_gemm.hip; K-heavy products of the plain loader) against the register-staged path it replaced,
    through every epilogue: the same MFMAs on the same operands in the same order, so the results must agree bit for bit -- including row / column
    tails (M, N not multiples of the tile) that the DMA path fills through the buffer descriptor's bounds check."""
    gen = torch.Generator().manual_seed(7 * M + N + K)
    a = (torch.randn(M, K, generator=gen) + torch.arange(M)[:, None] * 1e-4).to(dtype).cuda()
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dtype).cuda()
    b = torch.randn(N, generator=gen).cuda()
    prev = torch.randn(M, N, generator=gen).to(dtype).cuda()
    x = torch.randn(M, N, generator=gen).cuda()

    def run_all():
        outs = [ops.linear(a, w, b), ops.linear(a, w, b, 1), *ops.linear_pre_gelu(a, w, b), ops.linear_mul_dgelu(a, w, b, prev)]
        if M % 64 == 0:
            B, H = (M // 4096, 64) if M % 4096 == 0 else (M // 64, 8)
            sc = (1.25 * (torch.arange(B) % 3 != 0).float()).cuda()
            outs += [ops.linear_residual(a, w, b, x, sc, B, H, H), ops.linear_residual(a, w, b, x, sc, B, H, H, windowed=True, shift=4 if H > 8 else 0)]
        if N % 96 == 0 and K == N // 3 and M % 64 == 0:
            outs += list(ops.qkv(a, w, b, K // 32))
        return outs

    monkeypatch.setenv("UF_VARIANT", "gemm_dma=0")
    ref = run_all()
    monkeypatch.setenv("UF_VARIANT", "gemm_dma=1")
    got = run_all()
    torch.cuda.synchronize()
    for i, (g, r) in enumerate(zip(got, ref)):
        assert torch.equal(g, r), f"output {i} of {M}x{N}x{K} differs between the DMA and the register-staged path: max abs {(g.float() - r.float()).abs().max().item():.3e}"
    check(f"dma_linear_{M}x{N}x{K}", got[0], a.float().cpu() @ w.float().cpu().t() + b.cpu(), dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("C,heads", [(16, 1), (32, 1), (64, 2), (128, 4), (256, 8), (512, 16)])
def test_ln_fused_projections(ops, dtype, C, heads):
    """uf_ln_qkv_fwd / uf_ln_linear_gelu_fwd == the unfused kernels' composition == the oracle's LN + linear."""
    gen = torch.Generator().manual_seed(C + 1)
    B, H, W = 3, 16, 24            # 1152 tokens: not a multiple of the 128-row block (tail path)
    M = B * H * W
    x = torch.randn(M, C, generator=gen) * 1.5 + 0.3
    gm, bt = 1 + 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)
    mod = 0.5 * torch.randn(64, C, generator=gen)
    wqkv = (torch.randn(3 * C, C, generator=gen) / C ** 0.5).to(dtype)
    bqkv = 0.1 * torch.randn(3 * C, generator=gen)
    w1 = (torch.randn(4 * C, C, generator=gen) / C ** 0.5).to(dtype)
    b1 = 0.1 * torch.randn(4 * C, generator=gen)
    hd = C // heads
    for shift, m_ in ((0, None), (4, mod)):
        z = O.layer_norm(x, gm, bt).reshape(B, H, W, C)
        z = O.window_partition(torch.roll(z, shifts=(-shift, -shift), dims=(1, 2)), 8).reshape(-1, 64, C)
        if m_ is not None:
            z = z + m_
        zz = z.to(dtype).float() if dtype != torch.float32 else z
        y = zz.reshape(-1, C) @ wqkv.float().t() + bqkv
        nw = M // 64
        qr = (y[:, :C] * hd ** -0.5).reshape(nw, 64, heads, hd).permute(0, 2, 1, 3)
        kr = y[:, C:2 * C].reshape(nw, 64, heads, hd).permute(0, 2, 1, 3)
        vtr = y[:, 2 * C:].reshape(nw, 64, heads, hd).permute(0, 2, 3, 1)
        q, k, vt = ops.ln_qkv(x.cuda(), gm.cuda(), bt.cuda(), wqkv.cuda(), bqkv.cuda(), heads, B=B, H=H, W=W, shift=shift,
                              modulator=None if m_ is None else m_.cuda())
        check(f"ln_qkv_q_C{C}_s{shift}", q, qr, dtype)
        check(f"ln_qkv_k_C{C}_s{shift}", k, kr, dtype)
        check(f"ln_qkv_vt_C{C}_s{shift}", vt, vtr, dtype)
    z = O.layer_norm(x, gm, bt)
    zz = z.to(dtype).float() if dtype != torch.float32 else z
    ref = O.gelu_erf(zz @ w1.float().t() + b1)
    got = ops.ln_linear_gelu(x.cuda(), gm.cuda(), bt.cuda(), w1.cuda(), b1.cuda())
    check(f"ln_linear_gelu_C{C}", got, ref, dtype)


@pytest.mark.parametrize("dtype", MODES)
def test_window_attention_golden(ops, golden, dtype):
    """WindowAttention.forward fixture from the reference (C=64, heads=2, 8 windows of a 16x16 map)."""
    from uformer_amd import packing
    g = golden("window_attention")
    p = params(g, "p.")
    x = t(g["x"])
    a = x.reshape(-1, 64).to(dtype).cuda()
    wqkv = torch.cat([p["qkv.to_q.weight"], p["qkv.to_kv.weight"]], 0).to(dtype).cuda()
    bqkv = torch.cat([p["qkv.to_q.bias"], p["qkv.to_kv.bias"]], 0).cuda()
    bias = packing.rpb_dense(p["relative_position_bias_table"], p["relative_position_index"]).cuda()
    q, k, vt = ops.qkv(a, wqkv, bqkv, 2)
    for name, kw in (("nomask", dict(shift=0)), ("mask", dict(shift=0, mask=t(g["mask"]).cuda())),
                     ("mask", dict(shift=4))):       # dense mask and the analytic SW-MSA mask must agree
        o = ops.window_attention_core(q, k, vt, bias, H=16, W=16, **kw)
        y = ops.linear(o, p["proj.weight"].to(dtype).cuda(), p["proj.bias"].cuda(), 0)
        check(f"window_attention_{name}_{'analytic' if kw.get('shift') else 'dense'}", y.reshape(8, 64, 64),
              t(g["y_" + name]), dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("heads,hd", [(1, 32), (4, 32), (2, 16), (16, 32)])
def test_window_attention_oracle(ops, dtype, heads, hd):
    from uformer_amd import packing
    C = heads * hd
    gen = torch.Generator().manual_seed(C)
    H = W = 24
    nwin = 2 * 9
    x = torch.randn(nwin, 64, C, generator=gen)
    p = {"qkv.to_q.weight": torch.randn(C, C, generator=gen) / C ** 0.5 * 2, "qkv.to_q.bias": 0.1 * torch.randn(C, generator=gen),
         "qkv.to_kv.weight": torch.randn(2 * C, C, generator=gen) / C ** 0.5 * 2, "qkv.to_kv.bias": 0.1 * torch.randn(2 * C, generator=gen),
         "proj.weight": torch.eye(C), "proj.bias": torch.zeros(C),
         "relative_position_bias_table": torch.randn(225, heads, generator=gen),
         "relative_position_index": t(O.relative_position_index(8))}
    pd = {k: (v.to(dtype).float() if v.is_floating_point() and "weight" in k else v) for k, v in p.items()}
    xd = x.to(dtype).float()
    for shift in (0, 4):
        mask = O.shift_attn_mask(H, W, 8, 4) if shift else None
        ref = O.window_attention(xd, pd, "", heads, mask)
        a = x.reshape(-1, C).to(dtype).cuda()
        wqkv = torch.cat([p["qkv.to_q.weight"], p["qkv.to_kv.weight"]], 0).to(dtype).cuda()
        bqkv = torch.cat([p["qkv.to_q.bias"], p["qkv.to_kv.bias"]], 0).cuda()
        q, k, vt = ops.qkv(a, wqkv, bqkv, heads)
        o = ops.window_attention_core(q, k, vt, packing.rpb_dense(p["relative_position_bias_table"], p["relative_position_index"]).cuda(),
                                      H=H, W=W, shift=shift)
        check(f"attention_core_h{heads}_d{hd}_s{shift}", o.reshape(nwin, 64, C), ref, dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 8, 24, 128), (1, 40, 40, 16)])
def test_dwconv_gelu(ops, dtype, shape):
    from uformer_amd import packing
    B, H, W, C = shape
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(shape, generator=gen).to(dtype)
    w = torch.randn(C, 1, 3, 3, generator=gen) / 3
    b = 0.1 * torch.randn(C, generator=gen)
    ref = O.gelu_erf(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=1, groups=C)).permute(0, 2, 3, 1)
    got = ops.dwconv3x3_gelu(x.cuda(), packing.pack_dwconv(w).cuda(), b.cuda())
    check(f"dwconv_{H}x{W}x{C}", got, ref, dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("C,H,W", [(16, 8, 8), (32, 16, 24), (64, 16, 16), (128, 24, 8), (256, 16, 16), (512, 8, 16)])
def test_dwconv_linear2_fused(ops, dtype, C, H, W):
    """uf_dwconv_linear2_fwd == x + linear2(GELU(dwconv3x3(h1))) of the oracle (tile borders, image borders, B>1)."""
    from uformer_amd import packing
    gen = torch.Generator().manual_seed(C + H)
    B = 2
    hid = 4 * C
    h1 = torch.randn(B, H, W, hid, generator=gen).to(dtype)
    wd = torch.randn(hid, 1, 3, 3, generator=gen) / 3
    bd = 0.1 * torch.randn(hid, generator=gen)
    w2 = (torch.randn(C, hid, generator=gen) / hid ** 0.5).to(dtype)
    b2 = 0.1 * torch.randn(C, generator=gen)
    x = torch.randn(B * H * W, C, generator=gen)
    h2 = O.gelu_erf(torch.nn.functional.conv2d(h1.float().permute(0, 3, 1, 2), wd, bd, padding=1, groups=hid)).permute(0, 2, 3, 1)
    if dtype != torch.float32:
        h2 = h2.to(dtype).float()
    ref = x + h2.reshape(-1, hid) @ w2.float().t() + b2
    got = ops.dwconv_linear2(h1.cuda(), packing.pack_dwconv(wd).cuda(), bd.cuda(), w2.cuda(), b2.cuda(), x.cuda())
    check(f"dwconv_linear2_C{C}_{H}x{W}", got, ref, dtype)


@pytest.mark.parametrize("dtype", MODES)
def test_leff_golden(golden, dtype):
    from uformer_amd import model
    g = golden("leff")
    m = model.LeFF(16, 64)
    m.load_state_dict(params(g, "p."), strict=True)
    m = m.cuda().eval()
    check("leff", m(t(g["x"]).cuda(), compute_dtype=dtype), t(g["y"]), dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_lewin_block_golden(golden, dtype, tag):
    """LeWinTransformerBlock.forward fixtures from the reference: shifted/unshifted, +-modulator, user mask."""
    from uformer_amd import model
    g = golden("lewin_block_" + tag)
    heads = int(g["heads"])
    C = t(g["x"]).shape[-1]
    p = params(g, "p.")
    x = t(g["x"]).cuda()
    for shift in (0, 4):
        blk = model.LeWinTransformerBlock(C, (16, 16), heads, win_size=8, shift_size=shift, token_mlp="leff",
                                          modulator=(tag == "b"))
        blk.load_state_dict(p, strict=True)
        blk = blk.cuda().eval()
        check(f"lewin_block_{tag}_shift{shift}", blk(x, None, dtype), t(g[f"y_shift{shift}"]), dtype)
        if tag == "b":
            um = t(g["user_mask"]).cuda()
            check(f"lewin_block_{tag}_shift{shift}_usermask", blk(x[:1], um, dtype), t(g[f"y_shift{shift}_usermask"]), dtype)


def test_leff2_and_attn_block_variants_bit_identical():
    """The launch variants that are chosen by shape (and therefore by batch size) must agree bit for bit, or a batch-16 forward would not equal 16
    single forwards: leff2 with 8 producer waves / the persistent tile walk, attn_block in its low-register forms and (round 6) its
    single-operand-tile form at C = 256.  The
    switches are read once per process, so every variant runs in a child process and returns hashes of its outputs on the same inputs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from uformer_amd import model as um, ops
out = []
for (B, H, C) in ((1, 64, 128), (5, 64, 128), (1, 32, 256), (9, 32, 256), (1, 32, 512), (3, 64, 64), (2, 128, 32)):
    g = torch.Generator().manual_seed(B * 7 + H + C)
    h1 = torch.randn(B, H, H, 4 * C, generator=g).to(torch.bfloat16).cuda()
    w9 = (torch.randn(9, 4 * C, generator=g) * 0.2).cuda(); bd = (torch.randn(4 * C, generator=g) * 0.1).cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(torch.bfloat16).cuda(); b2 = torch.randn(C, generator=g).cuda()
    x = torch.randn(B * H * H, C, generator=g).cuda()
    y = ops.dwconv_linear2(h1, w9, bd, w2, b2, x)
    out.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
for (B, H, C, heads) in ((2, 64, 32, 1), (1, 128, 128, 4), (2, 32, 64, 2), (5, 64, 256, 8)):      # the last: C = 256 with 320 windows (> 256: the 4-wave forms)
    torch.manual_seed(C + H)
    blk = um.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
    xb = torch.randn(B, H * H, C, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        y = blk(xb, None, torch.bfloat16)
    out.append(hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16])
print("HASHES " + " ".join(out))
""" % root
    res = {}
    for tag, env in (("default", {}), ("leff2 8 producers", {"UF_VARIANT": "leff2=1"}), ("leff2 never 8 producers", {"UF_VARIANT": "leff2=2"}),
                     ("leff2 one tile per workgroup", {"UF_VARIANT": "persist=0"}),
                     ("leff2 tile walk everywhere", {"UF_VARIANT": "persist=1,leff2=2"}), ("attn_block first form", {"UF_VARIANT": "attn=0"}),
                     ("attn_block low-register form", {"UF_VARIANT": "attn=1"}), ("attn_block low-register code, first bounds", {"UF_VARIANT": "attn=2"}),
                     ("attn_block C = 256 single operand tile (O overwrites Xn)", {"UF_VARIANT": "attn=3"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("HASHES ")]
        assert r.returncode == 0 and line, (tag, r.stdout[-1500:], r.stderr[-1500:])
        res[tag] = line[0]
    for tag, v in res.items():
        assert v == res["default"], f"{tag}: outputs differ from the default variant\n{v}\n{res['default']}"


@pytest.mark.parametrize("dtype", MODES)
def test_samplers_golden(golden, dtype):
    from uformer_amd import model
    g = golden("samplers")
    dn = model.Downsample(8, 16); dn.load_state_dict(params(g, "dn.")); dn = dn.cuda()
    up = model.Upsample(16, 8); up.load_state_dict(params(g, "up.")); up = up.cuda()
    ip = model.InputProj(3, 8, 3, 1); ip.load_state_dict(params(g, "ip.")); ip = ip.cuda()
    op = model.OutputProj(16, 3, 3, 1); op.load_state_dict(params(g, "op.")); op = op.cuda()
    check("downsample", dn(t(g["xd"]).cuda(), dtype), t(g["yd"]), dtype)
    check("upsample", up(t(g["xu"]).cuda(), dtype), t(g["yu"]), dtype)
    check("input_proj", ip(t(g["xi"]).cuda()), t(g["yi"]), torch.float32)
    check("output_proj", op(t(g["xo"]).cuda()), t(g["yo"]), torch.float32)


def test_stem_beside_mfma_kernels_of_another_stream(ops, monkeypatch):
    """The stem on a side stream while the main stream runs kernels with MFMA waves (the GEMM, a LeWin block): the situation in which the first build of
    the LDS-staged stem returned wrong pixels in lanes 48-63 -- its packed FMAs selected the high half of a register pair (uf_elementwise.hip, kernel
    comment; scripts/ubench_hip/pk_opsel.hip).  Every side-stream result must equal the first form's, bit for bit."""
    from uformer_amd import model as um
    torch.manual_seed(0)
    B, H, E = 8, 256, 32
    img = torch.rand(B, 3, H, H, device="cuda")
    w27 = (torch.randn(27, E, device="cuda") * 0.2).contiguous()
    bias = (torch.randn(E, device="cuda") * 0.1).contiguous()
    monkeypatch.setenv("UF_VARIANT", "stem=1")
    ref = ops.input_proj(img, w27, bias)
    monkeypatch.setenv("UF_VARIANT", "stem=2")
    side = torch.cuda.Stream()
    ga = torch.randn(131072, 256, device="cuda").to(torch.bfloat16)
    gw = torch.randn(1024, 256, device="cuda").to(torch.bfloat16)
    gb = torch.zeros(1024, device="cuda")
    blk = um.LeWinTransformerBlock(32, (H, H), 1, win_size=8, shift_size=0, modulator=True).cuda().eval()
    xb = torch.randn(B, H * H, 32, device="cuda")

    def block():
        with torch.no_grad():
            blk(xb, None, torch.bfloat16)

    for name, fn in (("GEMM", lambda: ops.linear(ga, gw, gb)), ("LeWin block", block)):
        bad = 0
        for _ in range(20):
            for _ in range(3):
                fn()
            with torch.cuda.stream(side):
                y = ops.input_proj(img, w27, bias)
            torch.cuda.synchronize()
            bad += 0 if torch.equal(y, ref) else 1
        assert bad == 0, f"stem beside {name}: {bad} of 20 side-stream results differ from the first form"


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (1, 40, 72), (3, 8, 130), (1, 13, 70)])
def test_stem_head_lds_forms_bit_identical(ops, B, H, W, monkeypatch):
    """The LDS-staged stem / head kernels (round 4: input_proj2 with the image tile + weights in LDS, output_proj2 with the halo tile of
    token rows fetched once by LDS-DMA) against the first global-load forms they replace: same products, same order of additions ->
    bit-identical, including image borders and sizes that are not multiples of the tile; and against the oracle."""
    gen = torch.Generator().manual_seed(B * 1000 + H + W)
    img = torch.rand(B, 3, H, W, generator=gen).cuda()
    for E in (16, 32):
        w = (torch.randn(E, 3, 3, 3, generator=gen) * 0.2)
        bias = torch.randn(E, generator=gen) * 0.1
        from uformer_amd import packing
        w27 = packing.pack_input_proj(w).cuda()
        monkeypatch.setenv("UF_VARIANT", "stem=1")
        ref = ops.input_proj(img, w27, bias.cuda())
        monkeypatch.setenv("UF_VARIANT", "stem=2")                   # the default form
        got = ops.input_proj(img, w27, bias.cuda())
        assert torch.equal(got, ref), f"input_proj E={E}: LDS form differs, max abs {(got - ref).abs().max().item():.3e}"
        ora = O.input_proj(img.cpu(), {"input_proj.proj.0.weight": w, "input_proj.proj.0.bias": bias})
        check(f"input_proj2_E{E}_{H}x{W}", got.reshape(B, H * W, E), ora, torch.float32)
    for C2 in (16, 32, 64):
        x = torch.randn(B * H * W, C2, generator=gen).cuda()
        w = torch.randn(3, C2, 3, 3, generator=gen) * 0.1
        bias = torch.randn(3, generator=gen) * 0.1
        from uformer_amd import packing
        wp = packing.pack_output_proj(w).cuda()
        for add in (None, img):
            monkeypatch.setenv("UF_VARIANT", "head=1")
            ref = ops.output_proj(x, wp, bias.cuda(), B, H, W, add)
            monkeypatch.setenv("UF_VARIANT", "head=2")
            got = ops.output_proj(x, wp, bias.cuda(), B, H, W, add)
            assert torch.equal(got, ref), f"output_proj C2={C2}: LDS form differs, max abs {(got - ref).abs().max().item():.3e}"
        ora = torch.nn.functional.conv2d(x.cpu().reshape(B, H, W, C2).permute(0, 3, 1, 2), w, bias, stride=1, padding=1) + img.cpu()    # OutputProj + global residual (model.py:828-836, :1305)
        check(f"output_proj2_C{C2}_{H}x{W}", got, ora, torch.float32)


@pytest.mark.parametrize("dtype", MODES)
def test_samplers_oracle_bigger(dtype):
    """Downsample / Upsample at widths that cross GEMM tile edges (N=256/1024, K=2048/256)."""
    from uformer_amd import model
    gen = torch.Generator().manual_seed(3)
    dn = model.Downsample(128, 256)
    up = model.Upsample(256, 64)
    x = torch.randn(2, 24 * 24, 128, generator=gen)
    xu = torch.randn(2, 12 * 12, 256, generator=gen)
    pd = {"conv.0.weight": dn.conv[0].weight.detach().to(dtype).float(), "conv.0.bias": dn.conv[0].bias.detach()}
    pu = {"deconv.0.weight": up.deconv[0].weight.detach().to(dtype).float(), "deconv.0.bias": up.deconv[0].bias.detach()}
    check("downsample_128", dn.cuda()(x.cuda(), dtype), O.downsample(x.to(dtype).float(), pd, ""), dtype)
    check("upsample_256", up.cuda()(xu.cuda(), dtype), O.upsample(xu.to(dtype).float(), pu, ""), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,H,W,B", [(32, 32, 64, 2), (32, 256, 256, 1), (64, 16, 32, 3), (64, 128, 128, 1), (128, 8, 32, 2), (128, 64, 64, 1), (256, 8, 16, 2), (256, 32, 32, 3)])
def test_downsample_forms_bit_identical(ops, dtype, C, H, W, B, monkeypatch):
    """Downsample, second form (uf_gemm.hip down_patch_kernel: the input pixels under a tile of output pixels staged in LDS once, every operand fragment of every
    tap an LDS read) against the im2col-loader GEMM it replaces at C = 32 ... 256: same K order, same MFMAs -- bit for bit, on maps with image borders
    inside every tile (the smallest maps are one tile), several images, non-square maps, an input row stride wider than C; and both against the oracle
    (Conv2d k4 s2 p1, model.py:739-746)."""
    gen = torch.Generator().manual_seed(11 * C + H + W)
    ld = C + 8                                                     # rows wider than C: the loader must honour ld_x
    xw = torch.randn(B * H * W, ld, generator=gen).cuda()
    x = xw[:, :C]
    w4 = torch.randn(2 * C, C, 4, 4, generator=gen) / (16 * C) ** 0.5
    bias = torch.randn(2 * C, generator=gen).cuda()
    from uformer_amd import packing
    wp = packing.pack_downsample(w4.cuda(), dtype)

    def run():
        out = torch.empty(B * (H // 2) * (W // 2), 2 * C, dtype=torch.float32, device="cuda")
        lib = ops._lib.load()
        ops._lib.check(lib.uf_downsample_fwd(xw.data_ptr(), ld, wp.data_ptr(), bias.data_ptr(), out.data_ptr(), 2 * C, B, H, W, C, ops.uf_dtype(dtype),
                                             torch.cuda.current_stream().cuda_stream), "uf_downsample_fwd")
        return out

    def run_fm():
        out = torch.empty(B * (H // 2) * (W // 2), 2 * C, dtype=torch.float32, device="cuda")
        lib = ops._lib.load()
        wfm = ops.pack_weight_fm(wp)
        ops._lib.check(lib.uf_downsample_fm_fwd(xw.data_ptr(), ld, wp.data_ptr(), wfm.data_ptr(), bias.data_ptr(), out.data_ptr(), 2 * C, B, H, W, C, ops.uf_dtype(dtype),
                                                torch.cuda.current_stream().cuda_stream), "uf_downsample_fm_fwd")
        return out

    monkeypatch.setenv("UF_VARIANT", "down=1")
    ref = run()
    monkeypatch.setenv("UF_VARIANT", "down=2")
    got = run()
    monkeypatch.delenv("UF_VARIANT")
    got_fm = run_fm()                                             # the same form streaming the fragment-major pack of the weight (uf_downsample_fm_fwd)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), f"C={C} {H}x{W} B={B}: the two forms differ, max abs {(got - ref).abs().max().item():.3e}"
    assert torch.equal(got_fm, ref), f"C={C} {H}x{W} B={B}: the fragment-major weight stream differs, max abs {(got_fm - ref).abs().max().item():.3e}"
    xi = x.cpu().to(dtype).float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    ora = torch.nn.functional.conv2d(xi, w4.to(dtype).float(), bias.cpu(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 2 * C)
    check(f"downsample_patch_C{C}_{H}x{W}", got, ora, dtype)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("N,K", [(48, 16), (96, 32), (768, 256), (512, 2048)])
def test_fragment_major_packing_bit_exact(ops, dtype, N, K):
    """uf_pack_weight_fm (device) == packing.pack_frag (host formula) == the layout documented in the header."""
    from uformer_amd import packing
    w = torch.randn(N, K, generator=torch.Generator().manual_seed(N + K)).to(dtype)
    dev = ops._pack_frag(w.cuda()).cpu()
    host = packing.pack_frag(w, dtype).reshape(-1)
    assert torch.equal(dev, host)
    KS = (K + 31) // 32
    n, k = N - 3, K - 5                                    # spot-check the closed form
    off = ((n // 16 * KS + k // 32) * 64 + ((k % 32) // 8) * 16 + n % 16) * 8 + k % 8
    assert dev[off] == w[n, k]


def test_errors_are_loud(ops):
    """No silent fallbacks: CPU tensors, bad shapes and unsupported sizes raise with the library's message."""
    from uformer_amd._lib import UformerHipError
    with pytest.raises(UformerHipError, match="GPU only"):
        ops.window_partition(torch.zeros(1, 8, 8, 4), 8, 0)
    with pytest.raises(UformerHipError, match="multiple"):
        ops.linear(torch.zeros(64, 12, device="cuda", dtype=torch.bfloat16), torch.zeros(32, 12, device="cuda", dtype=torch.bfloat16),
                   torch.zeros(32, device="cuda"))
    with pytest.raises(UformerHipError, match="head_dim"):
        ops.qkv(torch.zeros(64, 48, device="cuda"), torch.zeros(144, 48, device="cuda"), torch.zeros(144, device="cuda"), 2)
    with pytest.raises(UformerHipError, match="unsupported"):
        ops.layernorm(torch.zeros(64, 48, device="cuda"), torch.ones(48, device="cuda"), torch.zeros(48, device="cuda"), B=1, H=8, W=8,
                      dtype=torch.float32)
