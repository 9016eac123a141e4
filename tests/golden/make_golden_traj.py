#!/usr/bin/env python3
"""K-step training TRAJECTORY fixture from the reference's own loop semantics (VERDICT r04 "next" 7; train/train_denoise.py:175-184:
``optimizer.zero_grad(); restored = model(input_); loss = criterion(restored, target); loss.backward(); optimizer.step()`` -- the
fp16 variant wraps the same sequence in autocast + loss_scaler).

The reference's ``Uformer`` (model.py, imported unmodified through the timm shim of make_golden.py) in train() mode, the reference's
``CharbonnierLoss`` (losses.py:41-52), ``torch.optim.AdamW(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)``
(train/train_denoise.py:77), tiny32 at 128x128, batch 2, 8 steps over two alternating batches; the DropPath masks the reference drew
are recorded per step.  Stored: the loss of every step, the output of the last step, and for every parameter signed projections of
(final - initial) plus a few whole tensors -- a step that used stale packed weights, a wrong bias-correction count or a wrong
DropPath stream cannot reproduce them.

Runs only in the build container (needs /root/reference):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_traj.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
import gradproj  # noqa: E402  (tests/gradproj.py: name-seeded signed projections)

ref, spec, save = mg.ref, mg.spec, mg.save
STEPS, LR, DROP_PATH = 8, 2e-4, 0.3


def main():
    torch.set_num_threads(8)
    sp = importlib.util.spec_from_file_location("ref_losses", os.path.join(mg.REF, "losses.py"))
    ref_losses = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(ref_losses)
    import timm.models.layers as tl
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
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
    torch.manual_seed(2024)
    m = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), win_size=8,
                    token_projection="linear", token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in, drop_path_rate=DROP_PATH).train()
    m.load_state_dict(sd, strict=True)
    init = {k: v.detach().clone() for k, v in m.named_parameters()}
    opt = torch.optim.AdamW(m.parameters(), lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)        # train/train_denoise.py:77
    crit = ref_losses.CharbonnierLoss()
    xs = [spec.synth_input(2, 128, 128, 9000 + i) for i in range(2)]
    ts = [spec.synth_input(2, 128, 128, 9100 + i) for i in range(2)]
    blocks = [b for b in m.modules() if isinstance(b, ref.LeWinTransformerBlock)]
    losses, step_masks = [], []
    for k in range(STEPS):
        masks.clear()
        opt.zero_grad()                                          # train/train_denoise.py:175
        restored = m(xs[k % 2])                                  # :181
        loss = crit(restored, ts[k % 2])                         # :182
        loss.backward()                                          # (:183 under the scaler)
        opt.step()                                               # :184
        losses.append(float(loss))
        full, it = [], iter(masks)
        for blk in blocks:                                       # nn.Identity where the rate is 0 (model.py:883): rows of ones
            for _ in range(2):
                full.append(next(it).clone() if isinstance(blk.drop_path, tl.DropPath) else torch.ones(2))
        assert next(it, None) is None
        step_masks.append(torch.stack(full))
        print(f"step {k}: loss {losses[-1]:.7f}")
    tl.DropPath.forward = orig_fwd
    out = {"losses": np.array(losses, dtype=np.float64), "masks": torch.stack(step_masks), "y_last": restored.detach(),
           "steps": STEPS, "lr": LR, "drop_path_rate": DROP_PATH,
           "drop_rates": np.array([float(b.drop_path.drop_prob) if isinstance(b.drop_path, tl.DropPath) else 0.0 for b in blocks]),
           "source": "reference model.py Uformer (train mode) + losses.py CharbonnierLoss + torch.optim.AdamW, the loop of train/train_denoise.py:175-184"}
    names, proj, dmax = [], [], []
    for k_, p_ in m.named_parameters():
        d = (p_.detach() - init[k_])
        names.append(k_)
        proj.append([float((d.double() * gradproj.proj_vector(k_, j, d.shape).double()).sum()) for j in range(2)])
        dmax.append(float(d.abs().max()))
        if k_ in ("input_proj.proj.0.weight", "encoderlayer_0.blocks.0.attn.qkv.to_q.weight", "conv.blocks.0.mlp.linear1.0.bias",
                  "decoderlayer_3.blocks.0.modulator.weight", "decoderlayer_0.blocks.0.attn.relative_position_bias_table",
                  "dowsample_1.conv.0.bias", "decoderlayer_2.blocks.0.norm2.weight", "upsample_3.deconv.0.weight"):
            out["final." + k_] = p_.detach()
            out["delta." + k_] = d
    out.update(param_names=np.array(names), delta_proj=np.array(proj, dtype=np.float64), delta_max=np.array(dmax))
    save("traj_tiny32_8steps", **out)


if __name__ == "__main__":
    main()
