"""GPU parity of the backward building blocks (row a15) through the C ABI, against the closed forms of
oracle/uformer_oracle_bwd.py -- which tests/test_oracle_golden.py pins to the reference's own autograd.

Tolerances: f32 2e-4 relative to the largest reference magnitude; bf16 operands 2.5e-2 (inputs/outputs rounded to 8 bits,
sums in f32); f16 operands (11 bits: 8x finer than bf16) a quarter of the bf16 tolerance everywhere (``pick``)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import uformer_oracle as O
from oracle import uformer_oracle_bwd as OB

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 2.5e-2, torch.float16: 2.5e-2 / 4}
MODES = [torch.float32, torch.bfloat16, torch.float16]
TAG = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}
F16_LOSS_SCALE = 65536.0       # torch.cuda.amp.GradScaler's initial scale (the reference trains under it, train/train_denoise.py:42,180-184)


# Whole-model gates of the 2-byte modes = about twice what the kernels measure (VERDICT r03 "next" 8; measured on MI355X, profiles/r03_parity_grad_*.json:
# gradients bf16 4.2e-2 / f16 5.1e-3 relative, restored images bf16 <= 2.1e-3): a 2x regression fails.
GRAD_RTOL_BF16, GRAD_RTOL_F16 = 6e-2, 1e-2
BF16_Y_TOL = 4e-3


def pick(dtype, f32, bf16, f16=None):
    """tolerance by operand type: f16 = a quarter of bf16's (8x finer rounding, 2x headroom) unless given, never under f32's"""
    return f32 if dtype == torch.float32 else (bf16 if dtype == torch.bfloat16 else (f16 if f16 is not None else max(f32, bf16 / 4)))


def rel(a, b):
    return (a.float().cpu() - b).abs().max().item() / max(1e-12, b.abs().max().item())


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dtype", MODES)
def test_gelu_bwd(dtype):
    from uformer_amd import ops
    a = (torch.randn(3, 50, 64, generator=g(1)) * 2).to(dtype)
    dy = torch.randn(3, 50, 64, generator=g(2)).to(dtype)
    ref = dy.float() * OB.gelu_erf_grad(a.float())
    got = ops.gelu_bwd(a.cuda(), dy.cuda())
    assert got.dtype == dtype and rel(got, ref) < TOL[dtype]


@pytest.mark.parametrize("C,rows", [(32, 1000), (64, 257), (256, 4096), (512, 130), (1024, 65)])
def test_layernorm_bwd(C, rows):
    from uformer_amd import ops
    x = torch.randn(rows, C, generator=g(3)) * 1.7 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=g(4))
    dy = torch.randn(rows, C, generator=g(5))
    rdx, rdg, rdb = OB.layer_norm_bwd(x, gamma, dy)
    dx, dg, db = ops.layernorm_bwd(x.cuda(), gamma.cuda(), dy.cuda())
    assert rel(dx, rdx) < 2e-4 and rel(dg, rdg) < 2e-4 and rel(db, rdb) < 2e-4
    dx2, dg2, db2 = ops.layernorm_bwd(x.cuda(), gamma.cuda(), dy.cuda())          # two-stage sums: bit-reproducible
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 8, 24, 128), (3, 32, 32, 32)])
def test_dwconv3x3_backward(dtype, B, H, W, C):
    """Input gradient = the forward stencil with flipped taps (gelu off, no bias); tap / bias gradients = uf_dwconv3x3_wgrad."""
    from uformer_amd import ops
    h = torch.randn(B, H, W, C, generator=g(6)).to(dtype)
    dc = torch.randn(B, H, W, C, generator=g(7)).to(dtype)
    w = torch.randn(C, 1, 3, 3, generator=g(8)) * 0.3
    bias = torch.randn(C, generator=g(9)) * 0.1
    w9 = w.reshape(C, 9).t().contiguous()                                             # tap-major, as packing.pack_dwconv
    # forward without activation == F.conv2d (checks the gelu = 0 path the backward reuses)
    ref_c = F.conv2d(h.float().permute(0, 3, 1, 2), w, bias, padding=1, groups=C).permute(0, 2, 3, 1)
    assert rel(ops.dwconv3x3(h.cuda(), w9.cuda(), bias.cuda(), gelu=False), ref_c) < TOL[dtype]
    rdh, rdw, rdb = OB.dwconv3x3_bwd(h.float(), w, dc.float())
    dh = ops.dwconv3x3(dc.cuda(), w9.flip(0).contiguous().cuda(), None, gelu=False)
    assert rel(dh, rdh) < TOL[dtype]
    dw9, db = ops.dwconv3x3_wgrad(h.cuda(), dc.cuda())
    assert rel(dw9, rdw.reshape(C, 9).t()) < pick(dtype, 2e-4, 2e-3)   # bf16: inputs exact in f32 sums
    assert rel(db, rdb) < pick(dtype, 2e-4, 2e-3)
    dw9b, dbb = ops.dwconv3x3_wgrad(h.cuda(), dc.cuda())
    assert torch.equal(dw9, dw9b) and torch.equal(db, dbb)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 8, 24, 128), (3, 32, 32, 32), (5, 4, 8, 2048), (2, 64, 64, 96), (70, 8, 8, 16)])
def test_dwconv3x3_fused_backward(dtype, B, H, W, C):
    """uf_dwconv3x3_bwd (one pass over dc; h1 recomputed from the pre-activation) against the two kernels it replaces: the input gradient
    bit-identical to uf_dwconv3x3_mul_dgelu (f32: to an ulp), the tap / bias gradients equal to uf_dwconv3x3_wgrad(GELU(pre), dc) up to summation order,
    and against the oracle's closed forms; run-to-run bit-identical (no atomics)."""
    from uformer_amd import ops
    pre = torch.randn(B, H, W, C, generator=g(40)).to(dtype).cuda()
    dc = torch.randn(B, H, W, C, generator=g(41)).to(dtype).cuda()
    w = torch.randn(C, 1, 3, 3, generator=g(42)) * 0.3
    w9 = w.reshape(C, 9).t().contiguous().cuda()
    flip = w9.flip(0).contiguous()
    h1 = ops.gelu(pre)                                                                # what the forward stored: T(GELU(pre as stored))
    da, dw9, db = ops.dwconv3x3_bwd(dc, flip, pre)
    da_2k = ops.dwconv3x3_mul_dgelu(dc, flip, pre)
    # 2-byte operands: bit-identical.  f32: GELU' is inlined into another instruction stream (fma contraction may differ by an ulp)
    same = torch.equal(da, da_2k) if dtype != torch.float32 else rel(da, da_2k.cpu()) < 1e-6
    assert same, f"da differs from uf_dwconv3x3_mul_dgelu: max {(da.float() - da_2k.float()).abs().max().item():.3e}"
    dw9_2k, db_2k = ops.dwconv3x3_wgrad(h1, dc)
    assert rel(dw9, dw9_2k.cpu()) < 2e-5, f"dw9 vs two-kernel form: {rel(dw9, dw9_2k.cpu()):.3e}"
    assert rel(db, db_2k.cpu()) < 2e-5, f"db vs two-kernel form: {rel(db, db_2k.cpu()):.3e}"
    _, rdw, rdb = OB.dwconv3x3_bwd(h1.float().cpu(), w, dc.float().cpu())
    assert rel(dw9, rdw.reshape(C, 9).t()) < pick(dtype, 2e-4, 2e-3) and rel(db, rdb) < pick(dtype, 2e-4, 2e-3)
    da2, dw9b, dbb = ops.dwconv3x3_bwd(dc, flip, pre)
    assert torch.equal(da, da2) and torch.equal(dw9, dw9b) and torch.equal(db, dbb)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv3x3_fused_backward_chunked(dtype):
    """Tensors of 4 GiB or more (32-bit byte offsets in the kernels) are walked in chunks of whole images with the tap / bias gradients added in chunk
    order (the one-pass form used to return UF_ERR_SHAPE where the two-kernel form worked).  The limit is lowered through
    UF_DWBWD_MAX_BYTES in a fresh process so that a small tensor takes the chunked path: input gradient bit-identical to the unchunked run, tap /
    bias gradients equal up to the summation order."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from uformer_amd import ops
dt = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1]]
g = torch.Generator().manual_seed(5)
pre = torch.randn(7, 16, 24, 64, generator=g).to(dt).cuda(); dc = torch.randn(7, 16, 24, 64, generator=g).to(dt).cuda()
flip = (torch.randn(9, 64, generator=g) * 0.3).cuda()
da, dw, db = ops.dwconv3x3_bwd(dc, flip, pre)
torch.save((da.cpu(), dw.cpu(), db.cpu()), sys.argv[2])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tag = "f32" if dtype == torch.float32 else "bf16"
    outs = []
    for lim in ("0", str(2 * 16 * 24 * 64 * (4 if dtype == torch.float32 else 2) + 1)):     # 0 = the real 4 GiB limit; then two images per chunk
        path = f"/tmp/dwbwd_{tag}_{lim}.pt"
        env = dict(os.environ)
        if lim != "0":
            env["UF_DWBWD_MAX_BYTES"] = lim
        subprocess.check_call([sys.executable, "-c", code, tag, path], env=env)
        outs.append(torch.load(path))
    (da0, dw0, db0), (da1, dw1, db1) = outs
    assert torch.equal(da0, da1)
    assert rel(dw1, dw0) < 2e-5 and rel(db1, db0) < 2e-5
    assert not torch.equal(dw0, torch.zeros_like(dw0))


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 24, 64), (1, 8, 12, 32), (3, 32, 32, 128)])
def test_dwconv_that_activates_its_input_equals_the_stencil_on_the_stored_activation(dtype, B, H, W, C):
    """uf_dwconv3x3_gelu_in_pre_gelu_fwd(a) == uf_dwconv3x3_pre_gelu_fwd(T(GELU(a))) bit for bit (walking kernel and the one-column form):
    the training forward stores linear1's pre-activation only."""
    from uformer_amd import ops
    a = (torch.randn(B, H, W, C, generator=g(150)) * 1.5).to(dtype).cuda()
    w9 = (torch.randn(9, C, generator=g(151)) * 0.3).cuda()
    bias = (torch.randn(C, generator=g(152)) * 0.1).cuda()
    c0, g0 = ops.dwconv3x3_pre_gelu(ops.gelu(a), w9, bias)
    c1, g1 = ops.dwconv3x3_pre_gelu(a, w9, bias, gelu_in=True)
    assert torch.equal(c0, c1) and torch.equal(g0, g1)
    assert not torch.equal(c1.float(), torch.zeros_like(c1.float()))


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("M,N,K", [(1000, 128, 32), (4096, 1024, 256), (333, 48, 16), (70, 64, 64), (20000, 96, 512),
                                   (1000, 256, 256), (777, 512, 768), (12345, 256, 512)])      # the last three and (4096, 1024, 256): 256 x 256 tiles
def test_linear_wgrad_and_input_grad(dtype, M, N, K, monkeypatch):
    """dW = dY^T X, db = column sums (token-split MFMA kernel, two-stage sums) and dX = dY W through the forward GEMM."""
    from uformer_amd import ops
    monkeypatch.setenv("UF_VARIANT", "wgrad4=1")           # the 256 x 256-tile kernel on every shape it supports (by default only with >= 64 K tokens)
    x = torch.randn(M, K, generator=g(10)).to(dtype)
    dy = torch.randn(M, N, generator=g(11)).to(dtype)
    w = (torch.randn(N, K, generator=g(12)) / K ** 0.5).to(dtype)
    rdx, rdw, rdb = OB.linear_bwd(x.float(), w.float(), dy.float())
    dW, db = ops.linear_wgrad(dy.cuda(), x.cuda())
    tol = pick(dtype, 2e-4, 2e-3)
    assert rel(dW, rdw) < tol and rel(db, rdb) < tol
    dW2, db2 = ops.linear_wgrad(dy.cuda(), x.cuda())
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    dx = ops.linear(dy.cuda(), w.t().contiguous().cuda(), torch.zeros(K).cuda())      # dX = dY W: the forward kernel, transposed weight
    assert rel(dx, rdx) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("v4", ["0", "1"])
@pytest.mark.parametrize("M,N,K", [(2049, 256, 256), (4113, 512, 256), (1000, 256, 512)])
def test_linear_wgrad_never_reads_past_the_last_token(dtype, v4, M, N, K, monkeypatch):
    """the LDS-DMA weight-gradient kernels (linear_wgrad3: 64-token steps, linear_wgrad4: 32-token stages) advance through the buffer
    instruction's SCALAR offset and rely on the descriptor's bounds check to zero the rows past M in the last step.  Here M is not a multiple of 64 and
    the rows directly behind dY and X (same allocation) are NaN: a tail that were read would poison dW / db."""
    from uformer_amd import ops
    monkeypatch.setenv("UF_VARIANT", "wgrad4=" + v4)
    G = 192                                                       # guard rows behind the operands
    xb = torch.full((M + G, K), float("nan")).to(dtype).cuda()
    dyb = torch.full((M + G, N), float("nan")).to(dtype).cuda()
    x = torch.randn(M, K, generator=g(21)).to(dtype)
    dy = torch.randn(M, N, generator=g(22)).to(dtype)
    xb[:M].copy_(x); dyb[:M].copy_(dy)
    dW, db = ops.linear_wgrad(dyb[:M], xb[:M])                    # contiguous row slices of the guarded allocations: same pointers, M rows
    assert torch.isfinite(dW).all() and torch.isfinite(db).all()
    rdw = dy.float().t() @ x.float()
    assert rel(dW, rdw) < 2e-3 and rel(db, dy.float().sum(0)) < 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(8192, 1024, 256), (5000, 256, 1024), (2049, 768, 256)])
def test_linear_wgrad_tile_versions_agree(dtype, M, N, K, monkeypatch):
    """The 256 x 256-tile kernel (uf_bwd.hip linear_wgrad4) against the 128 x 128-tile one on the same operands: both add exact products in f32,
    only the order of the partial sums differs -- and a strided dY (ld > N, as the fused qkv gradient is laid out) through the C ABI directly."""
    from uformer_amd import _lib, ops
    x = torch.randn(M, K, generator=g(13)).to(dtype).cuda()
    dy = torch.randn(M, N, generator=g(14)).to(dtype).cuda()
    monkeypatch.setenv("UF_VARIANT", "wgrad4=1")
    dW4, db4 = ops.linear_wgrad(dy, x)
    monkeypatch.setenv("UF_VARIANT", "wgrad4=0")
    dW3, db3 = ops.linear_wgrad(dy, x)
    monkeypatch.setenv("UF_VARIANT", "wgrad4=1")
    assert rel(dW4, dW3.cpu()) < 2e-6 and rel(db4, db3.cpu()) < 2e-6
    assert not torch.equal(dW3, torch.zeros_like(dW3))
    # strided operands: columns [N/2, N/2 + 256) of dy as a 256-wide layer's output gradient
    if N < 512:
        return
    lib = _lib.load()
    sub = dy[:, N // 2:N // 2 + 256]
    dWs = torch.empty(256, K, dtype=torch.float32, device="cuda")
    dbs = torch.empty(256, dtype=torch.float32, device="cuda")
    nbytes = lib.uf_linear_wgrad_workspace_bytes(M, 256, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    _lib.check(lib.uf_linear_wgrad(sub.data_ptr(), N, x.data_ptr(), K, dWs.data_ptr(), dbs.data_ptr(), M, 256, K, ops.uf_dtype(dtype), ws.data_ptr(), nbytes,
                                   torch.cuda.current_stream().cuda_stream), "uf_linear_wgrad")
    assert rel(dWs, dW3[N // 2:N // 2 + 256].cpu()) < 2e-6 and rel(dbs, db3[N // 2:N // 2 + 256].cpu()) < 2e-6


def _attention_bwd_reference(q, k, v, bias, mask, do, heads):
    """Closed form of the attention core's backward (the same algebra as OB.window_attention_bwd, without projections).
    q (scaled), k, v: (nW, heads, 64, hd); bias (heads,64,64); mask (nM,64,64) or None; do (nW, heads, 64, hd)."""
    s = q @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if mask is not None:
        nW, nM = s.shape[0], mask.shape[0]
        s = (s.reshape(nW // nM, nM, heads, 64, 64) + mask.unsqueeze(1).unsqueeze(0)).reshape(nW, heads, 64, 64)
    P = torch.softmax(s, dim=-1)
    dv = P.transpose(-2, -1) @ do
    dP = do @ v.transpose(-2, -1)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    return dS @ k, dS.transpose(-2, -1) @ q, dv, dS.sum(0)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("B,H,heads,shift", [(2, 16, 2, 4), (1, 32, 1, 0), (3, 16, 4, 4), (5, 8, 2, 0)])
def test_window_attention_bwd(dtype, B, H, heads, shift):
    from uformer_amd import ops
    nW = B * (H // 8) ** 2
    hd, C = 32, heads * 32
    q = (torch.randn(nW, heads, 64, hd, generator=g(20)) * hd ** -0.5).to(dtype)       # the forward stores q already scaled
    k = torch.randn(nW, heads, 64, hd, generator=g(21)).to(dtype)
    v = torch.randn(nW, heads, 64, hd, generator=g(22)).to(dtype)
    do = torch.randn(nW, heads, 64, hd, generator=g(23)).to(dtype)
    bias = torch.randn(heads, 64, 64, generator=g(24)) * 0.3
    mask = O.shift_attn_mask(H, H, 8, 4) if shift else None
    rdq, rdk, rdv, rdb = _attention_bwd_reference(q.float(), k.float(), v.float(), bias, mask, do.float(), heads)
    flat = lambda t: t.reshape(nW * heads, 64, hd).contiguous()                        # noqa: E731
    do_rows = do.permute(0, 2, 1, 3).reshape(nW * 64, C).contiguous()                  # merged heads, token rows
    dq, dk, dvt, dbias = ops.window_attention_bwd(flat(q).cuda(), flat(k).cuda(), flat(v).transpose(1, 2).contiguous().cuda(),
                                                  bias.cuda(), do_rows.cuda(), H, H, shift)       # analytic SW-MSA mask
    tol = TOL[dtype]
    assert rel(dq, flat(rdq)) < tol and rel(dk, flat(rdk)) < tol
    assert rel(dvt, flat(rdv).transpose(1, 2)) < tol
    assert rel(dbias, rdb) < tol
    if shift:   # the dense-mask argument must give the same result as the analytic mask
        dq2, dk2, dvt2, db2 = ops.window_attention_bwd(flat(q).cuda(), flat(k).cuda(), flat(v).transpose(1, 2).contiguous().cuda(),
                                                       bias.cuda(), do_rows.cuda(), H, H, 0, mask=mask.cuda())
        assert rel(dq2, flat(rdq)) < tol and rel(db2, rdb) < tol


@pytest.mark.parametrize("dtype", MODES)
def test_lewin_block_backward_vs_reference_autograd(golden, dtype):
    """A whole LeWin block (shifted windows, 2 heads, modulator): forward + backward assembled from the C-ABI kernels
    (uformer_amd/train.py) against the gradients the REFERENCE's autograd produced (tests/golden/grad_lewin_block.npz)."""
    import numpy as np
    from uformer_amd import train
    gd = golden("grad_lewin_block")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    p = {k[2:]: t(v).cuda() for k, v in gd.items() if k.startswith("p.")}
    y, dx, grads = train.lewin_block_forward_backward(t(gd["x"]).cuda(), p, "", int(gd["heads"]), 4, t(gd["gy"]).cuda(), dtype)
    tol = pick(dtype, 1e-3, 6e-2)
    assert rel(y, t(gd["y"])) < tol
    assert rel(dx, t(gd["dx"])) < tol, rel(dx, t(gd["dx"]))
    ref = {k[2:]: t(v) for k, v in gd.items() if k.startswith("g.")}
    assert set(grads) == set(ref), sorted(set(grads) ^ set(ref))
    worst = max((rel(grads[k], r), k) for k, r in ref.items())
    assert worst[0] < tol, worst


@pytest.mark.parametrize("dtype", MODES)
def test_model_backward_vs_reference_autograd(golden, dtype):
    """Whole tiny32 model (9 stages, samplers, stem, head, global residual) under the reference's Charbonnier loss: forward +
    backward assembled from the C-ABI kernels (uformer_amd/train.py) against the gradients the REFERENCE's autograd produced
    (tests/golden/grad_model_tiny32_128.npz): d loss / d input and all 299 parameter gradients."""
    import numpy as np
    from uformer_amd import spec, train
    gd = golden("grad_model_tiny32_128")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(1, 128, 128, 1234)
    target = spec.synth_input(1, 128, 128, 1235)
    # dL/dy of the Charbonnier loss (losses.py:41-52) at the REFERENCE forward output (= the oracle's, pinned to it): the loss
    # gradient is sign-like around |y - target| ~ eps, so it must not inherit the bf16 forward error of the path under test
    y_ref = O.uformer_forward(x, {k: v.cpu() for k, v in sd.items()}, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths,
                              num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    assert abs(O.charbonnier_loss(y_ref, target).item() - float(gd["loss"])) < 1e-6
    dy = OB.charbonnier_loss_bwd(y_ref, target)
    # f16 operands: scaled loss as under the reference's GradScaler (dy ~ 2e-5 here: f16 subnormals without it)
    y, dimg, grads = train.uformer_forward_backward(x.cuda(), sd, dy.cuda(), cfg=cfg, dtype=dtype, loss_scale=F16_LOSS_SCALE if dtype == torch.float16 else 1.0)
    assert rel(y, y_ref) < pick(dtype, 1e-5, 1e-2)
    tol = pick(dtype, 2e-3, 1e-1)
    assert rel(dimg, t(gd["dx"])) < tol, rel(dimg, t(gd["dx"]))
    names = [str(n) for n in gd["param_names"]]
    assert set(grads) == set(names)
    worst = (0.0, "")
    for n, (s_sum, s_abs, s_max) in zip(names, gd["grad_stats"]):
        e = abs(grads[n].abs().sum().item() - s_abs) / max(s_abs, 1e-12)
        worst = max(worst, (e, n))
    assert worst[0] < tol, worst
    for kname in [k for k in gd if k.startswith("g.")]:
        assert rel(grads[kname[2:]], t(gd[kname])) < tol, (kname, rel(grads[kname[2:]], t(gd[kname])))


@pytest.mark.parametrize("dtype", MODES)
def test_module_train_mode_loss_backward_with_droppath(golden, dtype):
    """The nn.Module boundary in train() mode: ``loss.backward()`` through uformer_amd.model.Uformer (UformerFunction over
    the C-ABI kernels) with the stochastic-depth masks the reference drew, against the reference's own train-mode forward
    and gradients (tests/golden/grad_model_tiny32_droppath.npz; train/train_denoise.py:180-184 is this sequence)."""
    import numpy as np
    from uformer_amd import model, spec
    gd = golden("grad_model_tiny32_droppath")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    cfg = spec.arch_config("tiny32", img_size=128)
    m = model.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                      modulator=cfg.modulator, dd_in=cfg.dd_in, drop_path_rate=0.5, compute_dtype=dtype)
    m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
    m = m.cuda().train()
    assert np.allclose(m.drop_path_rates(), [0.0] + list(gd["drop_rates"]) + [0.0], atol=1e-6)      # schedule of model.py:1093-1095
    m._drop_scales_override = t(gd["masks"]).cuda()
    x = spec.synth_input(2, 128, 128, 4321).cuda().requires_grad_(True)
    target = spec.synth_input(2, 128, 128, 4322).cuda()
    y = m(x)
    d = y - target
    loss = torch.mean(torch.sqrt(d * d + 1e-6))                             # CharbonnierLoss, eps = 1e-3 (losses.py:41-52)
    tolf = pick(dtype, 1e-5, 1e-2)
    assert rel(y.detach(), t(gd["y"])) < tolf and abs(loss.item() - float(gd["loss"])) < pick(dtype, 1e-5, 2e-3)
    ls = F16_LOSS_SCALE if dtype == torch.float16 else 1.0          # f16: scaled loss, as under the reference's GradScaler
    if dtype != torch.float32:       # the sign-like loss gradient must not inherit the 2-byte forward error: feed the reference's
        dref = t(gd["y"]).cuda() - target
        y.backward(dref / torch.sqrt(dref * dref + 1e-6) / dref.numel() * ls)
    else:
        loss.backward()
    if ls != 1.0:
        x.grad.div_(ls)
        for p_ in m.parameters():
            if p_.grad is not None:
                p_.grad.div_(ls)
    tol = pick(dtype, 2e-3, 1e-1)
    assert rel(x.grad, t(gd["dx"])) < tol
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n, (s_sum, s_abs, s_max) in zip([str(n) for n in gd["param_names"]], gd["grad_stats"]):
        gr = params[n].grad
        val = 0.0 if gr is None else gr.abs().sum().item()
        worst = max(worst, (abs(val - s_abs) / s_abs if s_abs > 0 else val, n))
    assert worst[0] < tol, worst
    for kname in [k for k in gd if k.startswith("g.")]:
        assert rel(params[kname[2:]].grad, t(gd[kname])) < tol, kname


def test_use_checkpoint_takes_the_recompute_form(golden):
    """``use_checkpoint=True`` (model.py:1056-1057: torch.utils.checkpoint around every block) selects the recompute form of the tape -- a block
    keeps only its input -- and its gradients agree with the kept-intermediates form of the same model on the same DropPath masks."""
    import numpy as np
    from uformer_amd import model, spec, train
    gd = golden("grad_model_tiny32_droppath")
    cfg = spec.arch_config("tiny32", img_size=128)
    x = spec.synth_input(2, 128, 128, 4321).cuda()
    target = spec.synth_input(2, 128, 128, 4322).cuda()
    grads, peak = {}, {}
    for ckpt in (False, True):
        m = model.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                          modulator=cfg.modulator, dd_in=cfg.dd_in, drop_path_rate=0.5, use_checkpoint=ckpt, compute_dtype=torch.bfloat16)
        m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
        m = m.cuda().train()
        m._drop_scales_override = torch.from_numpy(np.asarray(gd["masks"])).cuda()
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
        d = m(x) - target
        torch.mean(torch.sqrt(d * d + 1e-6)).backward()
        torch.cuda.synchronize()
        peak[ckpt] = torch.cuda.max_memory_allocated() - base
        assert train.UformerFunction.last_recompute is ckpt
        grads[ckpt] = {n: p_.grad.float().cpu() for n, p_ in m.named_parameters() if p_.grad is not None}
    assert grads[True].keys() == grads[False].keys()
    worst = max((rel(grads[True][n].cuda(), grads[False][n]), n) for n in grads[True] if grads[False][n].abs().max() > 0)
    assert worst[0] < 6e-2, worst           # both are bf16 roundings of the same gradient (GRAD_RTOL_BF16 against the reference)
    assert peak[True] < peak[False], peak   # the point of checkpointing


# ---- the separate GELU pass of the training forward (UF_TRAIN_SEPARATE_GELU, off by default until the training step is re-timed)
@pytest.mark.parametrize("dtype", MODES)
def test_gelu_fwd(dtype):
    from oracle import uformer_oracle as O
    from uformer_amd import ops
    a = (torch.randn(5, 33, 64) * 2).to(dtype)
    ref = O.gelu_erf(a.float())
    got = ops.gelu(a.cuda()).float().cpu()
    assert (got - ref).abs().max() < pick(dtype, 1e-6, 2.5e-2)


def test_recompute_backward_matches_stored_backward_and_new_reductions():
    """bf16: the fused-forward + recompute tape (one saved tensor per block) against the tape that stores every intermediate of
    the op-by-op forward -- same kernels in the backward, so the gradients agree to bf16 rounding of the forward difference; and
    the kernels that replaced ATen glue against torch: bias-table gather-sum vs index_add_, rows_sum vs sum(0), im2col/col2im
    vs unfold/fold."""
    import torch.nn.functional as F
    from uformer_amd import ops, spec, train
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(2, 128, 128, 3).cuda()
    dy = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(4)).cuda() * 1e-3
    drop = train.sample_drop_scales([0.3] * sum(cfg.depths), 2, "cuda", torch.Generator(device="cuda").manual_seed(5))
    y0, d0, g0 = train.uformer_forward_backward(x, sd, dy, cfg=cfg, dtype=torch.bfloat16, drop_scales=drop, recompute=False)
    y1, d1, g1 = train.uformer_forward_backward(x, sd, dy, cfg=cfg, dtype=torch.bfloat16, drop_scales=drop, recompute=True)
    assert rel(y1, y0.cpu()) < 2e-2 and rel(d1, d0.cpu()) < 5e-2
    assert set(g0) == set(g1)
    worst = max((rel(g1[k], g0[k].cpu()), k) for k in g0 if g0[k].abs().max() > 0)
    assert worst[0] < 8e-2, worst
    # bias-table gradient
    db = torch.randn(4, 64, 64, generator=torch.Generator().manual_seed(6))
    idx = spec.relative_position_index(8)
    ref = torch.zeros(225, 4).index_add_(0, idx.reshape(-1), db.permute(1, 2, 0).reshape(4096, 4))
    assert torch.allclose(ops.rpb_table_grad(db.cuda()).cpu(), ref, rtol=1e-5, atol=1e-6)
    # rows_sum
    for dt in (torch.float32, torch.bfloat16):
        xs = torch.randn(300, 64 * 32, generator=torch.Generator().manual_seed(7)).to(dt)
        assert torch.allclose(ops.rows_sum(xs.cuda()).cpu(), xs.float().sum(0), rtol=1e-4, atol=1e-3)
    # im2col / col2im, token layout (Downsample geometry) and NCHW (stem geometry)
    B, H, Cc = 2, 16, 8
    tok = torch.randn(B * H * H, Cc, generator=torch.Generator().manual_seed(8))
    img = tok.reshape(B, H, H, Cc).permute(0, 3, 1, 2).contiguous()
    for (k, s_, p_) in ((4, 2, 1), (3, 1, 1)):
        ref_cols = F.unfold(img, (k, k), padding=p_, stride=s_)                  # (B, Cin*k*k, P) with row index c*k*k + ky*k + kx
        P = ref_cols.shape[-1]
        ref_cols = ref_cols.reshape(B, Cc, k * k, P).permute(0, 3, 2, 1).reshape(B * P, k * k * Cc)     # -> column (ky,kx,c)
        cols = ops.im2col(tok.cuda(), B, H, H, Cc, k, s_, p_, torch.float32)
        assert torch.equal(cols.cpu()[:, :k * k * Cc], ref_cols)
        cols_n = ops.im2col(img.cuda(), B, H, H, Cc, k, s_, p_, torch.float32, nchw=True)
        assert torch.equal(cols_n.cpu(), cols.cpu())
        dcols = torch.randn(cols.shape, generator=torch.Generator().manual_seed(9))
        ref_dx = F.fold(dcols[:, :k * k * Cc].reshape(B, P, k * k, Cc).permute(0, 3, 2, 1).reshape(B, Cc * k * k, P), (H, H), (k, k), padding=p_, stride=s_)
        got = ops.col2im(dcols.cuda(), B, H, H, Cc, k, s_, p_)
        assert torch.allclose(got.cpu().reshape(B, H, H, Cc).permute(0, 3, 1, 2), ref_dx, rtol=1e-5, atol=1e-5)
        base = torch.ones(B * H * H, Cc).cuda()
        got2 = ops.col2im(dcols.cuda(), B, H, H, Cc, k, s_, p_, out=base)
        assert torch.allclose(got2, got + 1.0, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", MODES)
def test_uformer_B_256_backward_vs_reference_autograd(golden, dtype):
    """BASELINE configs[2]/[3] geometry: Uformer-B 256x256 under the reference's Charbonnier loss.  Forward + backward through
    the C-ABI kernels against the REFERENCE's autograd (tests/golden/grad_model_B_256.npz): loss, restored image, d loss /
    d input, and EVERY one of the 719 parameter gradients through two signed random projections and a seeded gather or the
    full tensor (tests/fixture_checks.py) -- a permuted or transposed gradient cannot pass.
    Tolerances: f32 2e-3 (of ||g|| / max|g|), bf16 6e-2 and f16 1e-2 (twice the measured errors, GRAD_RTOL_*); (round 3: f16 was bf16's / 4, derived from the
    mantissa widths) with the loss scaled by 65536 as under the reference's GradScaler -- d loss / d y is 5e-6 here, below f16's
    smallest normal number."""
    import fixture_checks as FC
    from uformer_amd import spec, train
    gd = golden("grad_model_B_256")
    cfg = spec.arch_config("Uformer_B", img_size=256)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(1, 256, 256, 1234)
    target = spec.synth_input(1, 256, 256, 1235)
    # dL/dy at the REFERENCE forward output (the oracle's, pinned to it by tests/test_oracle_golden.py): see the tiny32 test
    y_ref = O.uformer_forward(x, {k: v.cpu() for k, v in sd.items()}, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths,
                              num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    loss_ref = O.charbonnier_loss(y_ref, target).item()
    dy = OB.charbonnier_loss_bwd(y_ref, target)
    y, dimg, grads = train.uformer_forward_backward(x.cuda(), sd, dy.cuda(), cfg=cfg, dtype=dtype,
                                                    loss_scale=F16_LOSS_SCALE if dtype == torch.float16 else 1.0)
    # f16: the restored image meets the 1e-3 north-star tolerance like f32; gradients a quarter of the bf16 tolerance
    worst = FC.check_grad_B(gd, loss_ref, y, dimg, grads, rtol=pick(dtype, 2e-3, GRAD_RTOL_BF16, GRAD_RTOL_F16), loss_tol=1e-6, y_tol=BF16_Y_TOL if dtype == torch.bfloat16 else 1e-3)
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_grad_B_{TAG[dtype]}.json", "w") as f:
        json.dump(worst, f)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("M,N,K", [(1000, 256, 64), (4096, 128, 32), (130, 2048, 512)])
def test_fused_training_gemm_epilogues_equal_the_two_pass_forms(dtype, M, N, K):
    """uf_linear_pre_gelu_fwd == uf_linear_fwd then uf_gelu_fwd; uf_linear_mul_dgelu == uf_linear_fwd then uf_gelu_bwd: the fused
    stores apply the activation / its derivative to the value AS STORED, so the results are bit-identical to the two-pass forms."""
    from uformer_amd import ops
    a = torch.randn(M, K, generator=g(60)).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g(61)) * K ** -0.5).to(dtype).cuda()
    b = (torch.randn(N, generator=g(62)) * 0.1).cuda()
    pre0 = ops.linear(a, w, b)
    pre, act = ops.linear_pre_gelu(a, w, b)
    assert torch.equal(pre, pre0) and torch.equal(act, ops.gelu(pre0))
    c = (torch.randn(M, N, generator=g(63)) * 2).to(dtype).cuda()
    zero = torch.zeros(N, device="cuda")
    two_pass = ops.gelu_bwd(c, ops.linear(a, w, zero))
    assert rel(ops.linear_mul_dgelu(a, w, zero, c), two_pass.float().cpu()) < pick(dtype, 1e-6, 8e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cin,H,W,B,acc", [(32, 16, 32, 2, False), (32, 64, 64, 1, True), (64, 16, 32, 3, True), (128, 8, 32, 4, False), (128, 32, 64, 1, True), (256, 16, 32, 4, True), (256, 32, 32, 2, False)])
def test_downsample_input_gradient_patch_form(dtype, Cin, H, W, B, acc, monkeypatch):
    """uf_downsample_bwd's input gradient from an LDS patch of dy (down_dx_kernel: the four parity classes of ConvTranspose2d k4 s2 p1 as GEMMs over K = 4 taps x
    Cout, every tap's product rounded to T and the taps added in ascending (ky, kx)) against the patch-matrix route (GEMM rounded to T + col2im;
    UF_VARIANT="downdx=1"): BIT-IDENTICAL -- on maps that are one tile (every border inside it), several images, non-square maps, accumulation into an existing
    gradient -- and both against the oracle (conv_transpose2d of the T-rounded operands in f64).  The weight / bias gradients do not depend on the form."""
    from uformer_amd import ops, packing
    Cout = 2 * Cin
    x = torch.randn(B * H * W, Cin, generator=g(80 + Cin)).cuda()
    dy = torch.randn(B * (H // 2) * (W // 2), Cout, generator=g(81)).cuda()
    w4 = torch.randn(Cout, Cin, 4, 4, generator=g(82)) * (16 * Cin) ** -0.5
    wpt = packing.pack_downsample(w4.cuda(), dtype).t().contiguous()
    base = torch.randn(B * H * W, Cin, generator=g(83)).cuda()

    def run():
        add = base.clone() if acc else None
        dx, dW, db = ops.downsample_bwd(x, dy, wpt, B, H, W, add_to=add)
        return dx.clone(), dW.clone(), db.clone()

    monkeypatch.setenv("UF_VARIANT", "downdx=1")
    dx0, dW0, db0 = run()
    monkeypatch.delenv("UF_VARIANT")
    dx1, dW1, db1 = run()
    torch.cuda.synchronize()
    assert torch.equal(dW0, dW1) and torch.equal(db0, db1)
    dyq, wq = dy.cpu().to(dtype).double(), w4.to(dtype).double()
    ora = torch.nn.functional.conv_transpose2d(dyq.reshape(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2), wq, stride=2, padding=1)
    ora = ora.permute(0, 2, 3, 1).reshape(B * H * W, Cin).float() + (base.cpu() if acc else 0)
    scale = ora.abs().max().item()
    assert torch.equal(dx0, dx1), f"Cin={Cin} {H}x{W} B={B}: the two routes differ, max abs {(dx0 - dx1).abs().max().item():.3e}"
    assert (dx1.cpu() - ora).abs().max().item() / scale < pick(dtype, 1e-6, 8e-3)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 8, 24, 128), (3, 32, 32, 32)])
def test_fused_training_stencils_equal_the_two_pass_forms(dtype, B, H, W, C):
    from uformer_amd import ops
    h = torch.randn(B, H, W, C, generator=g(64)).to(dtype).cuda()
    a = (torch.randn(B, H, W, C, generator=g(65)) * 2).to(dtype).cuda()
    w9 = (torch.randn(9, C, generator=g(66)) * 0.3).cuda()
    bias = (torch.randn(C, generator=g(67)) * 0.1).cuda()
    pre0 = ops.dwconv3x3(h, w9, bias, gelu=False)
    pre, act = ops.dwconv3x3_pre_gelu(h, w9, bias)
    assert torch.equal(pre, pre0) and torch.equal(act, ops.gelu(pre0))
    flip = w9.flip(0).contiguous()
    # GELU' is inlined into a different instruction stream (fma contraction may differ by an ulp of the f32 product)
    two_pass = ops.gelu_bwd(a, ops.dwconv3x3(h, flip, None, gelu=False))
    assert rel(ops.dwconv3x3_mul_dgelu(h, flip, a), two_pass.float().cpu()) < pick(dtype, 1e-6, 8e-3)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("shift,windowed", [(0, False), (0, True), (4, True)])
def test_linear_residual_store(dtype, shift, windowed):
    """uf_linear_residual_fwd: x + scale[image] * (a W^T + b) with window_reverse / roll back in the GEMM's store, against the f32
    composition on the same (operand-rounded) inputs."""
    from uformer_amd import ops
    B, H, W, K, N = 3, 16, 24, 64, 32
    M = B * H * W
    a = torch.randn(M, K, generator=g(80)).to(dtype)
    w = (torch.randn(N, K, generator=g(81)) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g(82)) * 0.1
    x = torch.randn(M, N, generator=g(83))
    s = torch.tensor([0.0, 1.25, 1.25])
    branch = a.float() @ w.float().t() + b
    if windowed:
        branch = ops.window_reverse(branch.reshape(-1, 8, 8, N).cuda(), 8, H, W, shift).reshape(M, N).cpu()
    ref = x + branch * s.repeat_interleave(H * W).reshape(M, 1)
    out = ops.linear_residual(a.cuda(), w.cuda(), b.cuda(), x.cuda(), s.cuda(), B, H, W, windowed=windowed, shift=shift)
    assert rel(out, ref) < 2e-6
    out1 = ops.linear_residual(a.cuda(), w.cuda(), b.cuda(), x.cuda(), None, B, H, W, windowed=windowed, shift=shift)
    assert rel(out1, x + branch) < 2e-6                                            # no DropPath (eval, or rate 0)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("shift", [0, 4])
def test_backward_streaming_helpers(dtype, shift):
    """uf_residual_combine / uf_grad_fork / uf_qkv_grad_merge against the ATen sequences they replace (window_reverse + cast +
    per-sample scale + add;  add + scale + window_partition + cast;  head merge + query scale + cat)."""
    from uformer_amd import ops
    B, H, W, C, heads = 3, 16, 24, 64, 2
    M, hd = B * H * W, C // heads
    x = torch.randn(M, C, generator=g(70)).cuda()
    yw = torch.randn(M, C, generator=g(71)).to(dtype).cuda()                      # window order
    s = torch.tensor([0.0, 1.25, 1.25]).cuda()
    st = s.repeat_interleave(H * W).reshape(M, 1)
    ref = x + ops.window_reverse(yw.reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C).float() * st
    same = (lambda a, b: torch.equal(a, b)) if dtype != torch.float32 else (lambda a, b: rel(a, b.cpu()) < 1e-6)   # f32 b: the kernel's a + s*b is one fma
    assert same(ops.residual_combine(x, yw, s, B, H, W, windowed=True, shift=shift), ref)
    assert torch.equal(ops.residual_combine(None, yw, None, B, H, W, windowed=True, shift=shift),
                       ops.window_reverse(yw.reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C).float())
    assert same(ops.residual_combine(x, yw, s, B, H, W), x + yw.float() * st)
    g1, g2 = torch.randn(M, C, generator=g(72)).cuda(), torch.randn(M, C, generator=g(73)).cuda()
    tot, out = ops.grad_fork(g1, g2, s, B, H, W, dtype, windowed=True, shift=shift, want_sum=True)
    assert torch.equal(tot, g1 + g2)
    assert torch.equal(out, ops.window_partition(((g1 + g2) * st).reshape(B, H, W, C), 8, shift).reshape(M, C).to(dtype))
    none, out = ops.grad_fork(g1, None, None, B, H, W, dtype)
    assert none is None and torch.equal(out, g1.to(dtype))
    nW = M // 64
    dq = torch.randn(nW, heads, 64, hd, generator=g(74)).to(dtype).cuda()
    dk = torch.randn(nW, heads, 64, hd, generator=g(75)).to(dtype).cuda()
    dvt = torch.randn(nW, heads, hd, 64, generator=g(76)).to(dtype).cuda()
    merge = lambda t: t.reshape(nW, heads, 64, hd).permute(0, 2, 1, 3).reshape(M, C)      # noqa: E731
    ref = torch.cat([merge(dq.float() * hd ** -0.5).to(dtype), merge(dk), merge(dvt.transpose(2, 3))], 1)
    assert torch.equal(ops.qkv_grad_merge(dq, dk, dvt, heads), ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout,nchw", [(2, 24, 40, 3, 32, True), (1, 16, 16, 3, 16, True), (3, 8, 20, 4, 32, True), (1, 8, 12, 3, 64, True),
                                                  (2, 24, 40, 64, 3, False), (1, 16, 16, 32, 3, False), (5, 300, 12, 8, 4, False), (2, 12, 48, 64, 3, False),
                                                  (1, 9, 37, 3, 32, True)])
def test_conv3x3_bwd_direct_vs_torch_autograd(B, H, W, Cin, Cout, nchw):
    """uf_conv3x3_bwd (InputProj form with LeakyReLU' from the stored output, OutputProj form) against torch autograd on the CPU."""
    from uformer_amd import ops
    x = torch.randn(B, Cin, H, W, generator=g(80), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(81)) * 0.2).requires_grad_(True)
    b = (torch.randn(Cout, generator=g(82)) * 0.1).requires_grad_(True)
    dy = torch.randn(B, Cout, H, W, generator=g(83))
    pre = F.conv2d(x, w, b, padding=1)
    y = F.leaky_relu(pre, 0.01) if nchw else pre
    y.backward(dy)
    dy_rows = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda()
    if nchw:
        act = y.detach().permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda()
        dx, dW, db = ops.conv3x3_bwd(x.detach().cuda(), dy_rows, w.detach().cuda(), B, H, W, nchw=True, act_out=act, slope=0.01)
        ref_dx = x.grad
        none, dW2, db2 = ops.conv3x3_bwd(x.detach().cuda(), dy_rows, w.detach().cuda(), B, H, W, nchw=True, act_out=act, slope=0.01, need_dx=False)
        assert none is None and torch.equal(dW, dW2) and torch.equal(db, db2)         # fixed summation order
    else:
        rows = x.detach().permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
        dx, dW, db = ops.conv3x3_bwd(rows, dy_rows, w.detach().cuda(), B, H, W)
        ref_dx = x.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
    assert rel(dx, ref_dx) < 2e-5 and rel(dW, w.grad) < 2e-5 and rel(db, b.grad) < 2e-5


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("shift", [0, 4])
def test_window_attention_bwd_merged_output(dtype, shift):
    """uf_window_attention_bwd_qkv == uf_window_attention_bwd followed by the head merge (dq times head_dim^-0.5, scaled before the
    one rounding to T instead of after a first one)."""
    from uformer_amd import ops
    B, H, heads, hd = 2, 16, 4, 32
    C, nW = heads * hd, B * (H // 8) ** 2
    M = nW * 64
    q = (torch.randn(nW, heads, 64, hd, generator=g(90)) * hd ** -0.5).to(dtype).cuda()
    k = torch.randn(nW, heads, 64, hd, generator=g(91)).to(dtype).cuda()
    vt = torch.randn(nW, heads, hd, 64, generator=g(92)).to(dtype).cuda()
    bias = (torch.randn(heads, 64, 64, generator=g(93)) * 0.5).cuda()
    do = torch.randn(M, C, generator=g(94)).to(dtype).cuda()
    dq, dk, dvt, dbias = ops.window_attention_bwd(q, k, vt, bias, do, H, H, shift)
    dqkv, dbias2 = ops.window_attention_bwd_qkv(q, k, vt, bias, do, H, H, shift)
    merge = lambda t: t.reshape(nW, heads, 64, hd).permute(0, 2, 1, 3).reshape(M, C)      # noqa: E731
    assert torch.equal(dbias, dbias2)
    assert torch.equal(dqkv[:, C:2 * C], merge(dk)) and torch.equal(dqkv[:, 2 * C:], merge(dvt.transpose(2, 3)))
    assert rel(dqkv[:, :C], (merge(dq).float() * hd ** -0.5).cpu()) < pick(dtype, 1e-6, 8e-3)


@pytest.mark.parametrize("dtype", MODES)
def test_block_level_backward_entry_points(dtype):
    """uf_lewin_block_bwd (recomputation + backward in one C call) == uf_leff_bwd after uf_lewin_attn_bwd on the same operands == the
    op-by-op tape of uformer_amd/train.py (same kernels, same order: identical bits), with DropPath scales and a modulator."""
    from uformer_amd import ops, spec, train
    B, H, C, heads, shift = 3, 16, 64, 2, 4
    cfg = spec.arch_config("tiny32", 128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 11).items()}
    prefix = "decoderlayer_3.blocks.0."                                       # C = 64, 2 heads, modulator
    assert sd[prefix + "norm1.weight"].numel() == C and (prefix + "modulator.weight") in sd
    pk = train.BlockPack(sd, prefix, heads, shift, dtype, fused=False)
    x = torch.randn(B * H * H, C, generator=g(100)).cuda()
    dy = torch.randn(B * H * H, C, generator=g(101)).cuda()
    drop = torch.tensor([[1.25, 0.0, 1.25], [0.0, 1.25, 1.25]]).cuda()
    y, sv = train.lewin_block_forward(x.reshape(B, H * H, C), sd, prefix, heads, shift, dtype, drop, pk)
    dx_ref, g_ref = train.lewin_block_backward(sv, dy.reshape(B, H * H, C))
    dx, gv = ops.lewin_block_bwd(pk.train_params, x, dy, drop[0], drop[1], B, H, H, heads, dtype)
    named = train._named_block_grads(prefix, gv, C)
    assert torch.equal(dx, dx_ref.reshape(-1, C))
    assert set(named) == set(g_ref)
    for k in g_ref:
        assert torch.equal(named[k].reshape(g_ref[k].shape), g_ref[k]), k
    dx1, g_l = ops.lewin_block_bwd(pk.train_params, sv["x1"], dy, None, drop[1], B, H, H, heads, dtype, half="leff")
    dx2, g_a = ops.lewin_block_bwd(pk.train_params, x, dx1, drop[0], None, B, H, H, heads, dtype, half="attn")
    assert torch.equal(dx2, dx)
    for k in ("norm2_w", "norm2_b", "w1", "b1", "wdw", "bdw", "w2", "b2"):
        assert torch.equal(g_l[k], gv[k]), k
    for k in ("norm1_w", "norm1_b", "modulator", "rpb_table", "wqkv", "bqkv", "wproj", "bproj"):
        assert torch.equal(g_a[k], gv[k]), k


@pytest.mark.parametrize("dtype", MODES)
def test_native_block_pack_equals_the_aten_packing(dtype):
    """uf_pack_block_train (5 launches) against uformer_amd.packing / train.BlockPack (ATen casts, transposes, gathers): the fused
    forward and the block-level backward must give identical bits on either pack."""
    from uformer_amd import ops, spec, train
    B, H, C, heads, shift = 2, 16, 64, 2, 4
    cfg = spec.arch_config("tiny32", 128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 12).items()}
    prefix = "decoderlayer_3.blocks.0."
    py = train.BlockPack(sd, prefix, heads, shift, dtype, fused=True)
    nat = train.NativeBlockPack(sd, prefix, heads, shift, dtype)
    assert nat.fused.rpb_tab is not None                                      # the reference's index buffer: Toeplitz table in use
    x = torch.randn(B * H * H, C, generator=g(110)).cuda()
    dy = torch.randn(B * H * H, C, generator=g(111)).cuda()
    drop = torch.tensor([[1.25, 0.0], [1.25, 1.25]]).cuda()
    y1 = ops.lewin_block_train_fwd(py.fused, x, B, H, H, dtype, drop[0], drop[1])
    y2 = ops.lewin_block_train_fwd(nat.fused, x, B, H, H, dtype, drop[0], drop[1])
    assert torch.equal(y1, y2)
    dx1, g1 = ops.lewin_block_bwd(py.train_params, x, dy, drop[0], drop[1], B, H, H, heads, dtype)
    dx2, g2 = ops.lewin_block_bwd(nat.train_params, x, dy, drop[0], drop[1], B, H, H, heads, dtype)
    assert torch.equal(dx1, dx2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("C,shift", [(32, 0), (64, 4), (256, 4), (512, 0)])
def test_layernorm_bwd_fused_reads_window_order_and_adds_residual(dtype, C, shift):
    """uf_layernorm_bwd_fused == cast + window_reverse + uf_layernorm_bwd + add: dx to rounding (fma); dgamma / dbeta sum the rows in window
    order instead of raster order (rounding-level difference)."""
    from uformer_amd import ops
    B, H, W = 2, 16, 24
    M = B * H * W
    x = (torch.randn(M, C, generator=g(120)) * 1.5 + 0.2).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g(121))).cuda()
    dyw = torch.randn(M, C, generator=g(122)).to(dtype).cuda()                   # window order
    add = torch.randn(M, C, generator=g(123)).cuda()
    dy_raster = ops.window_reverse(dyw.reshape(-1, 8, 8, C), 8, H, W, shift).reshape(M, C).float()
    dx0, dg0, db0 = ops.layernorm_bwd(x, gamma, dy_raster)
    dx, dg, db = ops.layernorm_bwd_fused(x, gamma, dyw, B, H, W, add=add, windowed=True, shift=shift)
    assert rel(dx, (dx0 + add).cpu()) < 1e-6                                    # the kernel adds the residual gradient with one fma
    assert rel(dg, dg0.cpu()) < 1e-5 and rel(db, db0.cpu()) < 1e-5
    dx1, dg1, db1 = ops.layernorm_bwd_fused(x, gamma, dy_raster.to(dtype), B, H, W)    # raster order, no add: the plain form on T-typed dy
    dx2, dg2, db2 = ops.layernorm_bwd(x, gamma, dy_raster.to(dtype).float())
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)


@pytest.mark.parametrize("dtype", MODES)
@pytest.mark.parametrize("C", [32, 128, 512])
@pytest.mark.parametrize("ln_windowed,cast_windowed,shift", [(False, True, 0), (False, True, 4), (True, False, 4), (True, False, 0), (False, False, 0)])
def test_layernorm_bwd_cast_is_the_fork_of_its_dx(dtype, C, ln_windowed, cast_windowed, shift):
    """uf_layernorm_bwd_cast: dx, dgamma, dbeta bit-identical to uf_layernorm_bwd_fused, and the second output bit-identical to uf_grad_fork of
    that dx (per-image scales, window order with the cyclic shift): the two places the block backward uses it (LN2: token-order dy, windowed
    copy; LN1: window-order dy, token-order copy for the preceding block)."""
    from uformer_amd import ops
    B, H, W = 3, 16, 24
    M = B * H * W
    x = (torch.randn(M, C, generator=g(130)) * 1.5 + 0.2).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g(131))).cuda()
    dy = torch.randn(M, C, generator=g(132)).to(dtype).cuda()
    add = torch.randn(M, C, generator=g(133)).cuda()
    scale = torch.tensor([1.25, 0.0, 1.25]).cuda()
    for sc in (scale, None):
        dx0, dg0, db0 = ops.layernorm_bwd_fused(x, gamma, dy, B, H, W, add=add, windowed=ln_windowed, shift=shift)
        dx, dg, db, cast = ops.layernorm_bwd_fused(x, gamma, dy, B, H, W, add=add, windowed=ln_windowed, shift=shift,
                                                   cast=dict(scale=sc, windowed=cast_windowed, shift=shift))
        assert torch.equal(dx, dx0) and torch.equal(dg, dg0) and torch.equal(db, db0)
        _, ref = ops.grad_fork(dx0, None, sc, B, H, W, dtype, windowed=cast_windowed, shift=shift)
        assert cast.dtype == dtype and torch.equal(cast, ref)
        assert not torch.equal(cast.float(), torch.zeros_like(cast.float()))


def test_block_backward_with_and_without_the_fused_fork(monkeypatch):
    """The stored-intermediates backward of a stage of blocks with the operand copies written by the LayerNorm backward kernels against the
    same backward with separate uf_grad_fork passes: the fused kernel adds the residual gradient with one fma, so dx1 differs in the last f32
    bit here and there and a bf16 operand made from it by one ulp -- everything agrees to bf16 rounding."""
    from uformer_amd import spec, train
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 77).items()}
    x = spec.synth_input(2, 128, 128, 5).cuda()
    dy = torch.randn(2, 3, 128, 128, generator=g(140)).cuda() * 1e-3
    drop = train.sample_drop_scales([0.3] * sum(cfg.depths), 2, "cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(train, "_FUSE_FORK", fuse)
        outs.append(train.uformer_forward_backward(x, sd, dy, cfg=cfg, dtype=torch.bfloat16, drop_scales=drop, recompute=False))
    (y0, d0, g0), (y1, d1, g1) = outs
    assert torch.equal(y0, y1)
    assert rel(d0, d1.cpu()) < 2e-2                  # a last-bit difference of dx1 can move a bf16 operand by one ulp
    assert set(g0) == set(g1)
    worst = max((rel(g0[k], g1[k].cpu()), k) for k in g0 if g1[k].abs().max() > 0)
    assert worst[0] < 2e-2, worst


@pytest.mark.parametrize("dtype", MODES)
def test_uformer_T_head_dim16_trains_vs_reference_autograd(golden, dtype):
    """get_arch('Uformer_T') (embed_dim 16 -> head_dim 16 at every stage, utils/model_utils.py:66-67) trains: train() mode through the
    nn.Module boundary with the DropPath masks the reference drew, loss.backward(), against the REFERENCE's autograd
    (tests/golden/grad_model_T_128.npz: loss, restored images, d loss / d input, every parameter through signed probes).
    head_dim-16 blocks take the op-by-op forward + op-level backward with window_attn_bwd<16> (VERDICT r02 "missing" 2).
    Tolerances as the Uformer-B test: f32 2e-3, bf16 6e-2, f16 1e-2 (scaled loss)."""
    import fixture_checks as FC
    import numpy as np
    from uformer_amd import model, spec
    gd = golden("grad_model_T_128")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    m = model.get_arch("Uformer_T", 128, compute_dtype=dtype)
    cfg = spec.arch_config("Uformer_T", img_size=128)
    for blk, r in zip([b for n in spec.STAGES for b in getattr(m, n).blocks], gd["drop_rates"]):
        blk.drop_path_rate = float(r)                                       # the fixture was drawn at drop_path_rate 0.3 (make_golden_r3.py)
    m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
    m = m.cuda().train()
    m._drop_scales_override = t(gd["masks"]).cuda()
    x = spec.synth_input(2, 128, 128, 5321).cuda().requires_grad_(True)
    target = spec.synth_input(2, 128, 128, 5322).cuda()
    y = m(x)
    d = y - target
    loss = torch.mean(torch.sqrt(d * d + 1e-6))
    ls = F16_LOSS_SCALE if dtype == torch.float16 else 1.0
    if dtype != torch.float32:      # the sign-like loss gradient must not inherit the 2-byte forward error: feed the reference's
        dref = t(gd["y"]).cuda() - target
        y.backward(dref / torch.sqrt(dref * dref + 1e-6) / dref.numel() * ls)
    else:
        loss.backward()
    grads = {n: (None if p_.grad is None else p_.grad / ls) for n, p_ in m.named_parameters()}
    # the loss of THIS forward in every operand type (the 2-byte modes used to pass the golden value to its own check); its tolerance
    # follows the output tolerance: |d loss| <= mean |d y| for the Charbonnier loss
    worst = FC.check_grad_T(gd, loss.item(), y.detach(), x.grad / ls, grads, rtol=pick(dtype, 2e-3, GRAD_RTOL_BF16, GRAD_RTOL_F16), loss_tol=pick(dtype, 1e-5, 2e-3),
                            y_tol=BF16_Y_TOL if dtype == torch.bfloat16 else 1e-3)
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_grad_T_{TAG[dtype]}.json", "w") as f:
        json.dump(worst, f)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_standalone_block_module_is_differentiable(golden, dtype):
    """LeWinTransformerBlock as an ordinary differentiable nn.Module (the reference's is one, model.py:908-989; VERDICT r02 "missing"
    6): y = blk(x); y.backward(gy) in train() mode against the reference's autograd on the same block (tests/golden/grad_lewin_block.npz:
    shifted windows, 2 heads, modulator)."""
    import numpy as np
    from uformer_amd import model
    gd = golden("grad_lewin_block")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    C, heads = 64, int(gd["heads"])
    blk = model.LeWinTransformerBlock(C, (16, 16), heads, win_size=8, shift_size=4, token_mlp="leff", modulator=True)
    blk.load_state_dict({k[2:]: t(v) for k, v in gd.items() if k.startswith("p.")}, strict=True)
    blk = blk.cuda().train()
    x = t(gd["x"]).cuda().requires_grad_(True)
    y = blk(x, compute_dtype=dtype)
    assert y.grad_fn is not None
    ls = 1024.0 if dtype == torch.float16 else 1.0
    y.backward(t(gd["gy"]).cuda() * ls)
    tol = pick(dtype, 1e-3, 6e-2)
    assert rel(y.detach(), t(gd["y"])) < pick(dtype, 1e-5, 1e-2)
    assert rel(x.grad / ls, t(gd["dx"])) < tol
    for n, p_ in blk.named_parameters():
        assert rel(p_.grad / ls, t(gd["g." + n])) < tol, n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("prefix,C,heads,shift,B,H", [("encoderlayer_0.blocks.0.", 32, 1, 0, 2, 32), ("decoderlayer_3.blocks.0.", 64, 2, 4, 3, 16),
                                                      ("encoderlayer_2.blocks.1.", 128, 4, 4, 2, 16), ("decoderlayer_1.blocks.1.", 256, 8, 4, 1, 16),
                                                      ("decoderlayer_1.blocks.0.", 256, 8, 0, 5, 64), ("decoderlayer_0.blocks.1.", 512, 16, 4, 2, 8)])
def test_fused_attention_training_forward_equals_the_inference_kernel_and_the_op_by_op_forward(dtype, prefix, C, heads, shift, B, H):
    """uf_lewin_attn_train_fwd (round 6: the fused window kernel with side stores of what the backward reads): its residual rows are the bits of the
    inference kernel (uf_lewin_attn_fwd: same MFMAs on the same operands), and every stored operand -- xn, q, k, v^T, o (window rows), z, a1 (token
    rows) -- agrees with the op-by-op forward (layernorm -> qkv -> window_attention_core -> linear_residual -> layernorm -> linear) to the rounding of
    the operand type.  Widths 32 ... 512, both launch shapes of C = 256 (<= 256 and > 256 windows), shifted and unshifted, with and without modulator."""
    import ctypes
    from uformer_amd import _lib, ops, spec, train
    cfg = spec.arch_config("Uformer_B", 128)
    sd = {k: v.cuda() for k, v in spec.synth_state_dict(cfg, 21).items() if k.startswith(prefix)}
    assert sd[prefix + "norm1.weight"].numel() == C
    pk = train.NativeBlockPack(sd, prefix, heads, shift, dtype)
    x = (torch.randn(B * H * H, C, generator=g(200)) * 1.3).cuda()
    drop = (torch.rand(2, B, generator=g(201)) > 0.3).float().cuda() * 1.25
    M = B * H * H
    x1, xn, q, k, vt, o, z, a1 = ops.lewin_attn_train_fwd(pk.fused, x, B, H, H, heads, dtype, drop[0])
    # (1) residual rows without DropPath scales (uf_lewin_attn_fwd takes none): bit-identical to the inference kernel
    lib = _lib.load()
    xin = x.clone()
    dt = ops.uf_dtype(dtype)
    nbytes = lib.uf_block_workspace_bytes(M, C, dt)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    x1_nodrop, *_ = ops.lewin_attn_train_fwd(pk.fused, x, B, H, H, heads, dtype, None)
    _lib.check(lib.uf_lewin_attn_fwd(ctypes.byref(pk.fused), xin.data_ptr(), C, B, H, H, C, None, 0, dt, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "uf_lewin_attn_fwd")
    assert torch.equal(x1_nodrop, xin)
    assert torch.equal(x, (torch.randn(B * H * H, C, generator=g(200)) * 1.3).cuda())      # the input rows are untouched (out of place)
    # (2) against the op-by-op forward
    monkey = train._FUSED_ATTN_FWD
    train._FUSED_ATTN_FWD = False
    try:
        y_ref, sv = train.lewin_block_forward(x.reshape(B, H * H, C), sd, prefix, heads, shift, dtype, drop, pk)
    finally:
        train._FUSED_ATTN_FWD = monkey
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    for name, got in (("x1", x1), ("xn", xn), ("q", q), ("k", k), ("vt", vt), ("o", o), ("z", z), ("a1", a1)):
        ref = sv[name].float().cpu()
        assert got.shape == sv[name].shape and got.dtype == sv[name].dtype, name
        assert rel(got, ref) < tol, (name, rel(got, ref))
    # (3) the block forward built on it and its backward agree with the op-by-op pair to the same rounding
    y_f, sv_f = train.lewin_block_forward(x.reshape(B, H * H, C), sd, prefix, heads, shift, dtype, drop, pk)
    assert rel(y_f, y_ref.float().cpu()) < tol
    dy = torch.randn(B, H * H, C, generator=g(202)).cuda()
    dx_f, g_f = train.lewin_block_backward(sv_f, dy)
    dx_r, g_r = train.lewin_block_backward(sv, dy)
    assert rel(dx_f, dx_r.float().cpu()) < 2 * tol
    for name in g_r:
        if g_r[name].abs().max() > 0:
            assert rel(g_f[name], g_r[name].float().cpu()) < 2 * tol, name
