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


# ---- rows f-3 / f-4: the oracle helpers either side of the path, pinned to the reference's own code (tests/golden/make_golden_tail.py)
def test_tail_psnr_pinned_to_reference(golden):
    """O.psnr / O.batch_psnr vs the reference's myPSNR / batch_PSNR (utils/image_utils.py:40-51, ast-compiled)."""
    g = golden("tail_psnr")
    a, b = t(g["a"]), t(g["b"])
    for i in range(a.shape[0]):
        assert abs(O.psnr(a[i], b[i]) - float(g["per_image"][i])) < 1e-4
    assert abs(O.psnr(a, b) - float(g["whole"])) < 1e-4
    assert abs(O.batch_psnr(a, b, True) - float(g["avg"])) < 1e-4
    assert abs(O.batch_psnr(a, b, False) - float(g["total"])) < 1e-3


def test_tail_augment_and_mixup_pinned_to_reference(golden):
    """O.augment / O.crop_augment vs all 8 Augment_RGB_torch transforms, O.mixup vs MixUp_AUG.aug with its recorded draws
    (utils/dataset_utils.py:5-49, ast-compiled): bit-exact (pure index work; the mix is two f32 multiplies and an add)."""
    g = golden("tail_augment")
    x, frame = t(g["x"]), t(g["frame"])
    r, c, ps = int(g["crop_r"]), int(g["crop_c"]), int(g["crop_ps"])
    for k in range(8):
        assert torch.equal(O.augment(x, k), t(g[f"t{k}"])), k
        assert torch.equal(O.crop_augment(frame, r, c, ps, k), t(g[f"crop_t{k}"])), k
    perm, lam = t(g["mix_perm"]), t(g["mix_lam"])
    assert torch.equal(O.mixup(t(g["mix_gt"]), lam, perm), t(g["mix_gt_out"]))
    assert torch.equal(O.mixup(t(g["mix_noisy"]), lam, perm), t(g["mix_noisy_out"]))
    assert 0.0 < float(lam.min()) and float(lam.max()) < 1.0            # Beta(1.2, 1.2) draws


def test_tail_checkpoint_read_by_the_reference_loader(golden, tmp_path):
    """uformer_amd.checkpoint writes the dict the reference's load_checkpoint / load_optim / load_start_epoch read
    (utils/model_utils.py:23-54): the fixture holds the digests of what the REFERENCE's model and optimizer contained after loading
    the file this test rebuilds (same code, same seeds); plain and 'module.'-prefixed forms."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    g = golden("tail_checkpoint")
    # the fixture script's own helpers, minus everything that touches /root/reference
    import ast
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_tail.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("sd_digest", "optim_digest", "write_reference_style_checkpoint")]
    ns = {"torch": torch, "hashlib": hashlib, "spec": spec}
    exec(compile(ast.Module(body=body, type_ignores=[]), "make_golden_tail.py", "exec"), ns)
    from uformer_amd import checkpoint as ck
    from uformer_amd import model as um
    for prefix, tag in ((False, "plain."), (True, "dp.")):
        path = str(tmp_path / f"ck_{int(prefix)}.pth")
        cfg, ours, opt = ns["write_reference_style_checkpoint"](path, prefix)
        raw = torch.load(path, map_location="cpu")
        assert set(raw) == {"epoch", "state_dict", "optimizer"}                      # train/train_denoise.py:207-210
        assert all(k.startswith("module.") for k in raw["state_dict"]) == prefix
        assert ns["sd_digest"](ours.state_dict()) == str(g[tag + "state_digest"])    # what the reference's model held after load_checkpoint
        assert len(ours.state_dict()) == int(g[tag + "n_keys"])
        assert ck.load_start_epoch(path) == int(g[tag + "epoch"]) == 17
        m2 = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                        dd_in=cfg.dd_in, compute_dtype=torch.float32)
        ck.load_checkpoint(m2, path)
        assert ns["sd_digest"](m2.state_dict()) == str(g[tag + "state_digest"])
        o2 = torch.optim.AdamW(m2.parameters(), lr=9.9)
        assert abs(ck.load_optim(o2, path) - float(g[tag + "lr"])) < 1e-12
        assert ns["optim_digest"](o2.state_dict()) == str(g[tag + "optim_digest"])  # = the reference's optimizer after load_optim


def test_training_trajectory_oracle_loop_reproduces_reference(golden):
    """8 steps of the reference's loop (train/train_denoise.py:175-184: zero_grad / forward / CharbonnierLoss / backward / AdamW step; fixture from
    the reference's own model + loss + torch.optim.AdamW with recorded DropPath masks, tests/golden/make_golden_traj.py) replayed with the oracle
    forward under autograd: the loss of every step and the weight change of every parameter."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gradproj
    g = golden("traj_tiny32_8steps")
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=float(g["lr"]), betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    xs = [spec.synth_input(2, 128, 128, 9000 + i) for i in range(2)]
    ts = [spec.synth_input(2, 128, 128, 9100 + i) for i in range(2)]
    for k in range(int(g["steps"])):
        opt.zero_grad()
        loss = O.charbonnier_loss(O.uformer_forward(xs[k % 2], params, drop_scales=t(g["masks"][k]), **kw), ts[k % 2])
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - float(g["losses"][k])) < 1e-5 * float(g["losses"][k]), k
    for i, n in enumerate(str(n_) for n_ in g["param_names"]):
        d = params[n].detach() - sd[n]
        scale = float(g["delta_max"][i]) * d.numel() ** 0.5                 # size of a projection of an O(delta_max) tensor
        for j in range(2):
            pr = float((d.double() * gradproj.proj_vector(n, j, d.shape).double()).sum())
            assert abs(pr - float(g["delta_proj"][i][j])) < 2e-3 * scale + 1e-12, (n, j, pr, float(g["delta_proj"][i][j]))


def test_vendor_forward_equals_oracle():
    """oracle/vendor_forward.py (the reference's op sequence on torch's library ops; bench.py times it on the GPU box as the same-node
    vendor-stack calibration) is the same function as the oracle: tiny32 128x128, shifted and decoder (modulator) blocks included."""
    from oracle import vendor_forward as V
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
    x = spec.synth_input(2, 128, 128, 1234)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    with torch.no_grad():
        assert (O.uformer_forward(x, sd, **kw) - V.forward(x, sd, **kw)).abs().max().item() < 1e-5


def test_tail_ssim_against_reference_calculate_ssim(golden):
    """O.ssim vs the reference's OWN calculate_ssim / _ssim (utils/caculate_psnr_ssim.py:35-81), ast-compiled and run by
    tests/golden/make_golden_tail.py with a 2-function cv2 shim (getGaussianKernel, filter2D; cv2 is not installed in the build
    container) -- including a pair with out-of-range values, which the reference's uint8 conversion wraps (:59-62)."""
    g = golden("tail_ssim")
    assert str(g["pinned_by"]) == "reference"
    a, b = t(g["a"]), t(g["b"])
    for i in range(a.shape[0]):
        assert abs(O.ssim(a[i], b[i]) - float(g["ssim"][i])) < 1e-9
        assert abs(float(g["ssim_restated"][i]) - float(g["ssim"][i])) < 1e-9      # round 3's independent restatement agrees
    aw, bw = t(g["a_wrap"]), t(g["b_wrap"])
    for i in range(aw.shape[0]):
        assert abs(O.ssim(aw[i], bw[i]) - float(g["ssim_wrap"][i])) < 1e-9


def test_uformer_T_train_mode_every_parameter(golden):
    """get_arch('Uformer_T') (head_dim 16, utils/model_utils.py:66-67) in train() mode: the oracle with the recorded DropPath masks vs
    the reference's forward and autograd gradients (tests/golden/make_golden_r3.py), every parameter through the signed probes."""
    import fixture_checks as FC
    g = golden("grad_model_T_128")
    cfg = spec.arch_config("Uformer_T", img_size=128)
    assert list(g["heads"]) == list(cfg.num_heads) and all(cfg.embed_dim * m // h == 16 for m, h in zip((1, 2, 4, 8, 16, 16, 8, 4, 2), cfg.num_heads))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(2, 128, 128, 5321).requires_grad_(True)
    target = spec.synth_input(2, 128, 128, 5322)
    masks = t(g["masks"])
    assert masks.shape == (2 * sum(cfg.depths), 2) and (masks == 0).any()
    y = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in, drop_scales=masks)
    loss = O.charbonnier_loss(y, target)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point()}
    worst = FC.check_grad_T(g, loss.item(), y.detach(), x.grad, grads, rtol=GRAD_RTOL, loss_tol=1e-6, y_tol=5e-5)
    assert worst["proj"] < GRAD_RTOL
