#!/usr/bin/env python3
"""Golden GRADIENT fixtures, generated FROM THE REFERENCE ITSELF with torch autograd.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad.py

Same harness as make_golden.py (reference model.py imported unmodified behind the 3-symbol timm shim).  Modules are in
eval() mode, so DropPath is the identity and the gradients are deterministic; the loss of the whole-model fixture is the
reference's CharbonnierLoss (losses.py:41-52, eps 1e-3) against a seeded target, as in train/train_denoise.py:180-184.
These fixtures pin the backward of oracle/uformer_oracle.py (tests/test_oracle_golden.py) before any backward kernel
exists; they are the ground truth the round-2 backward kernels will be held to.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the shim, imports the reference as mg.ref)

ref, spec, g, save = mg.ref, mg.spec, mg.g, mg.save
sys.path.insert(0, mg.REF)
import losses as ref_losses  # noqa: E402  (reference losses.py: torch only)


def main():
    torch.set_num_threads(8)
    # ---------------- one LeWin block: shifted windows, 2 heads, modulator ---------------------------------
    C, heads = 64, 2
    blk = ref.LeWinTransformerBlock(C, (16, 16), heads, win_size=8, shift_size=4, token_mlp="leff", modulator=True).eval()
    mg.randomize_(blk, 131)
    x = torch.randn(2, 256, C, generator=g(132), requires_grad=True)
    gy = torch.randn(2, 256, C, generator=g(133))
    y = blk(x)
    y.backward(gy)
    save("grad_lewin_block", heads=heads, x=x.detach(), gy=gy, y=y.detach(), dx=x.grad,
         **{"p." + k: v for k, v in blk.state_dict().items()},
         **{"g." + k: p_.grad for k, p_ in blk.named_parameters()})

    # ---------------- per-op gradients: every backward kernel of the next round gets its own reference fixture -----------
    ops = {}

    def op_grads(tag, module, xin, gy_seed, **fwd_kw):
        """dx + all parameter gradients of one reference sub-module for a seeded upstream gradient."""
        xin = xin.clone().requires_grad_(True)
        out = module(xin, **fwd_kw)
        gy_ = torch.randn(out.shape, generator=g(gy_seed))
        out.backward(gy_)
        ops.update({f"{tag}.x": xin.detach(), f"{tag}.gy": gy_, f"{tag}.y": out.detach(), f"{tag}.dx": xin.grad})
        ops.update({f"{tag}.p.{k}": v for k, v in module.state_dict().items()})
        ops.update({f"{tag}.g.{k}": p_.grad for k, p_ in module.named_parameters()})

    attn = ref.WindowAttention(64, (8, 8), 2).eval(); mg.randomize_(attn, 211)
    blk16 = ref.LeWinTransformerBlock(32, (16, 16), 1, win_size=8, shift_size=4, token_mlp="leff")
    cap = {}
    orig = blk16.attn.forward
    blk16.attn.forward = lambda xx, attn_kv=None, mask=None: (cap.__setitem__("mask", mask.clone()), orig(xx, attn_kv, mask))[1]
    with torch.no_grad():
        blk16.eval()(torch.zeros(1, 256, 32))
    ops["attn.mask"] = cap["mask"]                                        # the SW-MSA mask of a 16x16 map (4 windows)
    op_grads("attn", attn, torch.randn(8, 64, 64, generator=g(212)), 213, mask=cap["mask"])
    lf = ref.LeFF(16, 64).eval(); mg.randomize_(lf, 221)
    op_grads("leff", lf, torch.randn(2, 256, 16, generator=g(222)), 223)
    dn = ref.Downsample(8, 16).eval(); mg.randomize_(dn, 231)
    op_grads("down", dn, torch.randn(2, 256, 8, generator=g(232)), 233)
    up = ref.Upsample(16, 8).eval(); mg.randomize_(up, 241)
    op_grads("up", up, torch.randn(2, 64, 16, generator=g(242)), 243)
    ip = ref.InputProj(3, 8, 3, 1, act_layer=torch.nn.LeakyReLU).eval(); mg.randomize_(ip, 251)
    op_grads("stem", ip, torch.rand(2, 3, 16, 16, generator=g(252)) - 0.3, 253)
    op = ref.OutputProj(16, 3, 3, 1).eval(); mg.randomize_(op, 261)
    op_grads("head", op, torch.randn(2, 256, 16, generator=g(262)), 263)
    ln = torch.nn.LayerNorm(48).eval()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(48, generator=g(271))); ln.bias.copy_(0.1 * torch.randn(48, generator=g(272)))
    op_grads("ln", ln, torch.randn(3, 20, 48, generator=g(273)) * 2 + 0.5, 274)
    save("grad_ops", **ops)

    # ---------------- whole tiny32 model, Charbonnier loss ---------------------------------------------------
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
    m = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                    win_size=8, token_projection="linear", token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in).eval()
    m.load_state_dict(sd, strict=True)
    xin = spec.synth_input(1, 128, 128, 1234).requires_grad_(True)
    target = spec.synth_input(1, 128, 128, 1235)
    loss = ref_losses.CharbonnierLoss()(m(xin), target)
    loss.backward()
    names = [k for k, _ in m.named_parameters()]
    stats = np.stack([[float(p_.grad.sum()), float(p_.grad.abs().sum()), float(p_.grad.abs().max())] for _, p_ in m.named_parameters()])
    full = {}
    for k, p_ in m.named_parameters():   # a representative of every parameter kind, stored in full
        if k in ("input_proj.proj.0.weight", "output_proj.proj.0.weight", "dowsample_0.conv.0.weight", "upsample_3.deconv.0.weight",
                 "encoderlayer_0.blocks.0.attn.qkv.to_q.weight", "encoderlayer_0.blocks.0.attn.relative_position_bias_table",
                 "encoderlayer_0.blocks.0.norm1.weight", "conv.blocks.0.mlp.dwconv.0.weight", "conv.blocks.0.mlp.linear1.0.bias",
                 "decoderlayer_3.blocks.0.modulator.weight", "decoderlayer_3.blocks.0.attn.qkv.to_kv.weight",
                 "decoderlayer_3.blocks.0.mlp.linear2.0.weight"):
            full["g." + k] = p_.grad
    assert len(full) == 12, sorted(full)
    save("grad_model_tiny32_128", loss=loss.detach(), dx=xin.grad, param_names=np.array(names), grad_stats=stats, **full)
    print("loss %.6f  |dx| max %.3e  params %d" % (float(loss), float(xin.grad.abs().max()), len(names)))

    # ---------------- train() mode: DropPath active (model.py:1093-1095 schedule), masks recorded ---------------------------
    # timm's DropPath scales a residual branch per sample by bernoulli(keep)/keep.  The masks the reference drew are recorded
    # in call order (two per block: attention branch, then LeFF branch, blocks in execution order) so that the oracle and the
    # HIP path can replay exactly the same stochastic depth.
    import timm.models.layers as tl
    masks = []
    orig_fwd = tl.DropPath.forward

    def recording_forward(self, x):
        if self.drop_prob == 0. or not self.training:
            masks.append(torch.ones(x.shape[0]))
            return x
        keep = 1 - self.drop_prob
        r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        r.div_(keep)
        masks.append(r.reshape(-1).clone())
        return x * r

    tl.DropPath.forward = recording_forward
    torch.manual_seed(77)
    mt = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                     win_size=8, token_projection="linear", token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in,
                     drop_path_rate=0.5).train()
    mt.load_state_dict(sd, strict=True)
    xin2 = spec.synth_input(2, 128, 128, 4321).requires_grad_(True)
    target2 = spec.synth_input(2, 128, 128, 4322)
    out2 = mt(xin2)
    loss2 = ref_losses.CharbonnierLoss()(out2, target2)
    loss2.backward()
    tl.DropPath.forward = orig_fwd
    # blocks whose rate is 0 hold nn.Identity instead of DropPath (model.py:883): give them rows of ones, so that the
    # stored array has two rows for EVERY block in execution order
    full, it = [], iter(masks)
    for blk in [m_ for m_ in mt.modules() if isinstance(m_, ref.LeWinTransformerBlock)]:
        for _ in range(2):
            full.append(next(it) if isinstance(blk.drop_path, tl.DropPath) else torch.ones(xin2.shape[0]))
    assert next(it, None) is None
    masks = full
    stats2 = np.stack([[float(p_.grad.sum()), float(p_.grad.abs().sum()), float(p_.grad.abs().max())] for _, p_ in mt.named_parameters()])
    dp_rates = [float(m.drop_prob) for m in mt.modules() if isinstance(m, tl.DropPath)]
    save("grad_model_tiny32_droppath", loss=loss2.detach(), y=out2.detach(), dx=xin2.grad, masks=torch.stack(masks), drop_rates=np.array(dp_rates),
         param_names=np.array([k for k, _ in mt.named_parameters()]), grad_stats=stats2,
         **{"g." + k: p_.grad for k, p_ in mt.named_parameters() if k in (
             "encoderlayer_0.blocks.0.attn.qkv.to_q.weight", "conv.blocks.0.mlp.linear1.0.bias", "decoderlayer_3.blocks.0.modulator.weight",
             "decoderlayer_3.blocks.0.mlp.linear2.0.weight", "dowsample_0.conv.0.weight", "input_proj.proj.0.weight")})
    print("train-mode loss %.6f  masks %s  dropped branches %d of %d" % (float(loss2), tuple(torch.stack(masks).shape),
          int((torch.stack(masks) == 0).sum()), torch.stack(masks).numel()))


if __name__ == "__main__":
    main()
