#!/usr/bin/env python3
"""Round-3 golden fixture, generated FROM THE REFERENCE ITSELF (VERDICT r02 "missing" 2 / "next" 6):

  * ``grad_model_T_128`` -- ``get_arch('Uformer_T')`` (utils/model_utils.py:66-67: embed_dim 16 -> head_dim 16 at every stage,
    depths [2]*9, drop_path_rate 0.1) in train() mode on two 128x128 images under the reference's ``CharbonnierLoss``
    (losses.py:41-52), gradients from the reference's autograd.  The DropPath masks the reference drew are recorded (two rows
    per block in execution order, as make_golden_grad.py does) so the path under test replays the same stochastic depth.
    Stored: loss, the restored images, d loss / d input, and for EVERY parameter two signed random projections, a seeded
    4096-element gather (or the full tensor) and 64x64 blocks of one parameter per kind and stage (tests/gradproj.py).

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (timm shim + the reference's model.py as mg.ref)

ref, spec, save = mg.ref, mg.spec, mg.save
sys.path.insert(0, os.path.dirname(HERE))
from gradproj import gather_index, proj_vector  # noqa: E402
sys.path.insert(0, mg.REF)
import losses as ref_losses  # noqa: E402


def param_probe(named_params):
    names = [k for k, _ in named_params]
    proj = np.zeros((len(names), 2), dtype=np.float64)
    norms = np.zeros((len(names), 2), dtype=np.float64)
    out, seen_kind = {}, set()
    for i, (k, p_) in enumerate(named_params):
        gr = p_.grad.detach()
        for j in range(2):
            proj[i, j] = float((gr.double() * proj_vector(k, j, gr.shape).double()).sum())
        norms[i] = (float(gr.double().pow(2).sum().sqrt()), float(gr.abs().max()))
        if gr.numel() <= 4096:
            out["full." + k] = gr
        else:
            out["gather." + k] = gr.reshape(-1)[gather_index(k, gr.numel())]
        parts = k.split(".")
        kind = (parts[0], ".".join(parts[3:])) if parts[1] == "blocks" else (parts[0], "")
        if kind not in seen_kind and gr.numel() > 4096:
            seen_kind.add(kind)
            out["block64." + k] = gr.reshape(gr.shape[0], -1)[:64, :64].clone()
    return names, proj, norms, out


def make_grad_T():
    import timm.models.layers as tl
    cfg = spec.arch_config("Uformer_T", img_size=128)
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
    torch.manual_seed(99)
    # exactly the kwargs of get_arch('Uformer_T') (utils/model_utils.py:66-67) + a drop rate that makes the masks bite on 2 samples
    m = ref.Uformer(img_size=128, embed_dim=16, win_size=8, token_projection="linear", token_mlp="leff", modulator=True, drop_path_rate=0.3).train()
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in sd.values()]
    m.load_state_dict(sd, strict=True)
    xin = spec.synth_input(2, 128, 128, 5321).requires_grad_(True)
    target = spec.synth_input(2, 128, 128, 5322)
    y = m(xin)
    loss = ref_losses.CharbonnierLoss()(y, target)
    loss.backward()
    tl.DropPath.forward = orig_fwd
    full, it = [], iter(masks)
    for blk in [m_ for m_ in m.modules() if isinstance(m_, ref.LeWinTransformerBlock)]:
        for _ in range(2):
            full.append(next(it) if isinstance(blk.drop_path, tl.DropPath) else torch.ones(xin.shape[0]))
    assert next(it, None) is None
    masks = torch.stack(full)
    rates = [float(b.drop_path.drop_prob) if isinstance(b.drop_path, tl.DropPath) else 0.0 for b in m.modules() if isinstance(b, ref.LeWinTransformerBlock)]
    names, proj, norms, out = param_probe(list(m.named_parameters()))
    save("grad_model_T_128", loss=loss.detach().double(), y=y.detach(), dx=xin.grad, masks=masks, drop_rates=np.array(rates), param_names=np.array(names),
         proj=proj, norms=norms, heads=np.array(cfg.num_heads), sd_sha256=mg.sd_digest(sd), **out)
    print(f"Uformer_T train-mode loss {float(loss):.6f}; masks {tuple(masks.shape)}, {int((masks == 0).sum())} dropped branch-samples; {len(names)} parameters, "
          f"{sum(1 for k in out if k.startswith('block64.'))} 64x64 blocks")


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_grad_T()
