"""Pin oracle/uformer_oracle.py against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Index ops: bit-exact.  Float: <= 2e-5 abs."""
import hashlib

import numpy as np
import torch

from oracle import uformer_oracle as O
from uformer_amd import spec

TOL = 2e-5


def t(a):
    return torch.from_numpy(np.asarray(a))


def params(g, prefix):
    return {k[len(prefix):]: t(v) for k, v in g.items() if k.startswith(prefix)}


def test_index_ops_bit_exact(golden):
    g = golden("index_ops")
    x = t(g["x"])
    wp = O.window_partition(x, 8)
    assert torch.equal(wp, t(g["partition"]))
    assert torch.equal(O.window_reverse(wp, 8, 16, 24), x)
    assert np.array_equal(O.window_partition_index(2, 16, 16, 8, 4), g["shifted_index_16"])
    assert np.array_equal(O.relative_position_index(8), g["rel_index"])
    assert torch.equal(spec.relative_position_index(8), t(g["rel_index"]))
    assert torch.equal(O.shift_attn_mask(16, 16, 8, 4), t(g["shift_mask_16"]))
    assert torch.equal(O.shift_attn_mask(32, 32, 8, 4), t(g["shift_mask_32"]))


def test_window_attention(golden):
    g = golden("window_attention")
    p = params(g, "p.")
    x = t(g["x"])
    y = O.window_attention(x, p, "", 2, None)
    assert (y - t(g["y_nomask"])).abs().max() < TOL
    y = O.window_attention(x, p, "", 2, t(g["mask"]))
    assert (y - t(g["y_mask"])).abs().max() < TOL


def test_leff(golden):
    g = golden("leff")
    y = O.leff(t(g["x"]), params(g, "p."), "")
    assert (y - t(g["y"])).abs().max() < TOL


def test_lewin_block(golden):
    for tag in ("a", "b"):
        g = golden("lewin_block_" + tag)
        p = params(g, "p.")
        heads = int(g["heads"])
        x = t(g["x"])
        for shift in (0, 4):
            y = O.lewin_block(x, p, "", heads, shift)
            assert (y - t(g[f"y_shift{shift}"])).abs().max() < TOL, (tag, shift)
        if tag == "b":
            um = t(g["user_mask"])
            for shift in (0, 4):
                y = O.lewin_block(x[:1], p, "", heads, shift, mask=um)
                assert (y - t(g[f"y_shift{shift}_usermask"])).abs().max() < TOL


def test_samplers(golden):
    g = golden("samplers")
    assert (O.downsample(t(g["xd"]), params(g, "dn."), "") - t(g["yd"])).abs().max() < TOL
    assert (O.upsample(t(g["xu"]), params(g, "up."), "") - t(g["yu"])).abs().max() < TOL
    ip = {"input_proj." + k: v for k, v in params(g, "ip.").items()}
    op = {"output_proj." + k: v for k, v in params(g, "op.").items()}
    assert (O.input_proj(t(g["xi"]), ip) - t(g["yi"])).abs().max() < TOL
    assert (O.output_proj(t(g["xo"]), op) - t(g["yo"])).abs().max() < TOL


def _sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest()


def _run_model(golden, tag):
    g = golden("model_" + tag)
    cfg = spec.arch_config(str(g["arch"]), img_size=int(g["img_size"]))
    sd = spec.synth_state_dict(cfg, int(g["seed"]))
    # the synthetic weights / inputs must be bit-identical to what the reference was fed
    assert _sd_digest(sd) == str(g["sd_sha256"])
    x = spec.synth_input(int(g["B"]), int(g["HW"]), int(g["HW"]), int(g["in_seed"]))
    assert hashlib.sha256(x.numpy().tobytes()).hexdigest() == str(g["x_sha256"])
    assert len(sd) == int(g["n_keys"])
    y = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths,
                          num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    return (y - t(g["y"])).abs().max().item()


def test_model_tiny(golden):
    assert _run_model(golden, "tiny_128") < TOL
    assert _run_model(golden, "tiny32_128") < TOL


def test_model_B(golden):
    assert _run_model(golden, "B_256") < 5e-5
    # constructor img_size=128 fed 256x256 input: bottleneck block 1 is NOT shifted (Appendix A-1)
    assert _run_model(golden, "B_ctor128_in256") < 5e-5


def test_block_shifts_ctor_clamp():
    assert spec.arch_config("Uformer_B", 256).block_shifts()[4] == [0, 4]
    assert spec.arch_config("Uformer_B", 128).block_shifts()[4] == [0, 0]
    assert O.block_shifts(128, (1, 2, 8, 8, 2, 8, 8, 2, 1))[4] == [0, 0]
    assert len(spec.state_dict_spec(spec.arch_config("Uformer_B"))) == 759


# ---- backward: the oracle is a functional torch restatement, so torch autograd differentiates it;
# these tests hold THAT backward to gradients produced by the reference's own autograd (tests/golden/make_golden_grad.py)
# -- the ground truth for the backward kernels of the next round.
GRAD_RTOL = 2e-4


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


def test_lewin_block_backward(golden):
    g = golden("grad_lewin_block")
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in params(g, "p.").items()}
    x = t(g["x"]).clone().requires_grad_(True)
    y = O.lewin_block(x, p, "", int(g["heads"]), 4)
    assert (y - t(g["y"])).abs().max() < TOL
    y.backward(t(g["gy"]))
    assert _rel(x.grad, t(g["dx"])) < GRAD_RTOL
    grads = params(g, "g.")
    assert len(grads) == 18      # every parameter of the block (the int64 index buffer has no gradient)
    for k, ref in grads.items():
        assert p[k].grad is not None, k
        assert _rel(p[k].grad, ref) < GRAD_RTOL, (k, _rel(p[k].grad, ref))


def test_model_backward_charbonnier(golden):
    g = golden("grad_model_tiny32_128")
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(1, 128, 128, 1234).requires_grad_(True)
    target = spec.synth_input(1, 128, 128, 1235)
    y = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    loss = O.charbonnier_loss(y, target)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    loss.backward()
    assert _rel(x.grad, t(g["dx"])) < GRAD_RTOL
    names = [str(n) for n in g["param_names"]]
    stats = g["grad_stats"]
    assert names == [k for k, v in sd.items() if v.is_floating_point()]       # parameter order == reference named_parameters()
    for n, (s_sum, s_abs, s_max) in zip(names, stats):
        gr = sd[n].grad
        assert gr is not None, n
        assert abs(gr.abs().sum().item() - s_abs) <= 5e-4 * s_abs + 1e-9, n
        assert abs(gr.abs().max().item() - s_max) <= 5e-4 * s_max + 1e-9, n
    full = params(g, "g.")
    assert len(full) == 12
    for k, ref in full.items():
        assert _rel(sd[k].grad, ref) < GRAD_RTOL, (k, _rel(sd[k].grad, ref))


def test_explicit_backward_block(golden):
    """oracle/uformer_oracle_bwd.py (closed-form backward, no autograd) vs the reference's autograd gradients."""
    from oracle import uformer_oracle_bwd as OB
    g = golden("grad_lewin_block")
    p = params(g, "p.")
    dx, grads = OB.lewin_block_bwd(t(g["x"]), p, "", int(g["heads"]), 4, t(g["gy"]))
    assert _rel(dx, t(g["dx"])) < GRAD_RTOL
    ref = params(g, "g.")
    assert set(grads) == set(ref)
    for k, r in ref.items():
        assert _rel(grads[k], r) < GRAD_RTOL, (k, _rel(grads[k], r))


def test_explicit_backward_model(golden):
    from oracle import uformer_oracle_bwd as OB
    g = golden("grad_model_tiny32_128")
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
    x = spec.synth_input(1, 128, 128, 1234)
    target = spec.synth_input(1, 128, 128, 1235)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    y = O.uformer_forward(x, sd, **kw)
    dx, grads = OB.uformer_backward(x, sd, OB.charbonnier_loss_bwd(y, target), **kw)
    assert _rel(dx, t(g["dx"])) < GRAD_RTOL
    names = [str(n) for n in g["param_names"]]
    assert set(grads) == set(names)
    for n, (s_sum, s_abs, s_max) in zip(names, g["grad_stats"]):
        assert abs(grads[n].abs().sum().item() - s_abs) <= 5e-4 * s_abs + 1e-9, n
        assert abs(grads[n].abs().max().item() - s_max) <= 5e-4 * s_max + 1e-9, n
    for k, r in params(g, "g.").items():
        assert _rel(grads[k], r) < GRAD_RTOL, (k, _rel(grads[k], r))


def test_explicit_backward_per_op(golden):
    """Every op-level closed form of oracle/uformer_oracle_bwd.py against the reference's autograd on that op alone
    (tests/golden/grad_ops.npz): the fixtures the individual backward kernels will be tested against."""
    from oracle import uformer_oracle_bwd as OB
    g = golden("grad_ops")

    def op(tag):
        pre = tag + "."
        return (t(g[pre + "x"]), t(g[pre + "gy"]), t(g[pre + "dx"]), params(g, pre + "p."), params(g, pre + "g."))

    def check(tag, dx, grads, ref_dx, ref_g, strip=""):
        assert _rel(dx, ref_dx) < GRAD_RTOL, (tag, _rel(dx, ref_dx))
        got = {k[len(strip):]: v for k, v in grads.items()}
        assert set(got) == set(ref_g), (tag, sorted(set(got) ^ set(ref_g)))
        for k, r in ref_g.items():
            assert _rel(got[k], r) < GRAD_RTOL, (tag, k, _rel(got[k], r))

    x, gy, rdx, p, rg = op("attn")
    check("attn", *OB.window_attention_bwd(x, p, "", 2, t(g["attn.mask"]), gy), rdx, rg)
    x, gy, rdx, p, rg = op("leff")
    check("leff", *OB.leff_bwd(x, p, "", gy), rdx, rg)
    x, gy, rdx, p, rg = op("down")
    check("down", *OB.downsample_bwd(x, p, "", gy), rdx, rg)
    x, gy, rdx, p, rg = op("up")
    check("up", *OB.upsample_bwd(x, p, "", gy), rdx, rg)
    x, gy, rdx, p, rg = op("stem")
    check("stem", *OB.input_proj_bwd(x, {"input_proj." + k: v for k, v in p.items()}, gy), rdx, rg, strip="input_proj.")
    x, gy, rdx, p, rg = op("head")
    check("head", *OB.output_proj_bwd(x, {"output_proj." + k: v for k, v in p.items()}, gy), rdx, rg, strip="output_proj.")
    x, gy, rdx, p, rg = op("ln")
    dx, dw, db = OB.layer_norm_bwd(x, p["weight"], gy)
    check("ln", dx, {"weight": dw, "bias": db}, rdx, rg)


def test_train_mode_droppath_forward_and_backward(golden):
    """train() mode: the oracle with the DropPath masks the reference drew (recorded in call order) reproduces the reference's
    stochastic-depth forward and, through autograd, its gradients."""
    g = golden("grad_model_tiny32_droppath")
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(2, 128, 128, 4321).requires_grad_(True)
    target = spec.synth_input(2, 128, 128, 4322)
    masks = t(g["masks"])
    assert masks.shape == (2 * sum(cfg.depths), 2) and (masks == 0).any()
    y = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads,
                          dd_in=cfg.dd_in, drop_scales=masks)
    assert (y - t(g["y"])).abs().max() < TOL
    loss = O.charbonnier_loss(y, target)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    loss.backward()
    assert _rel(x.grad, t(g["dx"])) < GRAD_RTOL
    for n, (s_sum, s_abs, s_max) in zip([str(n) for n in g["param_names"]], g["grad_stats"]):
        gr = sd[n].grad
        val = 0.0 if gr is None else gr.abs().sum().item()            # a block whose two branches were dropped for both samples has no gradient
        assert abs(val - s_abs) <= 5e-4 * s_abs + 1e-9, n
    for k, r in params(g, "g.").items():
        assert _rel(sd[k].grad, r) < GRAD_RTOL, k


# ---- round-2 fixtures at the BASELINE.json sizes (tests/golden/make_golden_r2.py) -----------------------------------------
def test_model_B_256_backward_charbonnier_every_parameter(golden):
    """Uformer-B 256x256 (BASELINE configs[2]/[3] geometry): autograd through the oracle vs the reference's autograd, every one of
    the 719 parameters through two signed random projections + a seeded gather (tests/fixture_checks.py)."""
    import fixture_checks as FC
    g = golden("grad_model_B_256")
    cfg = spec.arch_config("Uformer_B", img_size=256)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(1, 256, 256, 1234).requires_grad_(True)
    target = spec.synth_input(1, 256, 256, 1235)
    y = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    loss = O.charbonnier_loss(y, target)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point()}
    worst = FC.check_grad_B(g, loss.item(), y.detach(), x.grad, grads, rtol=GRAD_RTOL, loss_tol=1e-6, y_tol=5e-5)
    assert worst["proj"] < GRAD_RTOL


def test_model_B_720p_oracle_opt_in(golden):
    """BASELINE configs[4] through the ORACLE (about two CPU-minutes on 8 cores): opt in with UF_SLOW_TESTS=1.  The GPU test
    (tests/test_gpu_model.py::test_hires_720p_padded_to_1280) compares the HIP path with this same reference fixture directly."""
    import os

    import pytest
    if os.environ.get("UF_SLOW_TESTS") != "1":
        pytest.skip("opt-in (UF_SLOW_TESTS=1): 1280x1280 oracle forward takes minutes of CPU")
    import fixture_checks as FC
    g = golden("model_B_720p")
    img = spec.synth_input(1, 720, 1280, 9)
    xp, msk = O.expand2square(img, 128.0)
    assert hashlib.sha256(xp.numpy().tobytes()).hexdigest() == str(g["x_sha256"])     # the reference's expand2square, bit for bit
    cfg = spec.arch_config("Uformer_B", img_size=256)
    sd = spec.synth_state_dict(cfg, 1234)
    with torch.no_grad():
        y = O.uformer_forward(xp, sd, img_size=256, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    frame = torch.masked_select(y, msk.bool()).reshape(1, 3, 720, 1280)
    FC.check_720p(g, frame, 256, 1e-4, 2e-5)


def test_expand2square_matches_reference_digest(golden):
    """oracle.expand2square == the reference's own function (test/test_sidd.py:79-92) on the 720p frame: SHA-256 of the padded
    square stored by make_golden_r2.py, which compiled the reference's function from its source text."""
    g = golden("model_B_720p")
    xp, msk = O.expand2square(spec.synth_input(1, 720, 1280, 9), 128.0)
    assert hashlib.sha256(xp.numpy().tobytes()).hexdigest() == str(g["x_sha256"])
    assert float(msk.sum()) == float(g["mask_sum"]) == 720 * 1280
