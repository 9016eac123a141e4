#!/usr/bin/env python3
"""Round-2 golden fixtures, generated FROM THE REFERENCE ITSELF at the BASELINE.json sizes that had no reference
comparison in round 1 (VERDICT r01 "What's weak" 2 and 3):

  * ``model_B_720p``  -- BASELINE configs[4]: Uformer-B on a 1280x720 frame, padded by the reference's own
    ``expand2square`` (extracted from test/test_sidd.py:79-92 with ``ast``, so the script's argparse / dataset code
    does not run) to 1280x1280, forwarded through the unmodified reference ``Uformer`` built with ``img_size=256`` and
    with ``img_size=128`` (what the eval scripts do), cropped back with ``masked_select``
    (test/test_sidd.py:106-109).  Stored per constructor size: five 64x64 crops of the restored frame, a 16x16
    mean-pooled map of the whole frame, per-channel sum / abs-sum / max.
  * ``grad_model_B_256`` -- BASELINE configs[2]/[3] geometry: Uformer-B 256x256, B = 1, loss = the reference's
    ``CharbonnierLoss`` (losses.py:41-52), gradients from the reference's autograd in eval() mode.  Stored: loss, the
    restored image, d loss / d input (crop + pooled map), and for EVERY parameter two signed random projections
    ``(g * r_k).sum()`` with r_k ~ N(0,1) seeded by (name, k) plus a 4096-element seeded gather (or the whole tensor when
    it has <= 4096 elements); one parameter of every kind per stage additionally stores its top-left 64x64 block.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py [720p] [grad]
"""
import ast
import hashlib
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the timm shim, imports the reference as mg.ref)

ref, spec, save = mg.ref, mg.spec, mg.save
sys.path.insert(0, os.path.dirname(HERE))
from gradproj import gather_index, proj_vector  # noqa: E402  (tests/gradproj.py: shared with the tests)

CROPS_720P = {"centre": (328, 608), "top_left": (0, 0), "bottom_right": (656, 1216), "top_mid": (0, 608), "left_mid": (328, 0)}


def reference_expand2square():
    """The function object of test/test_sidd.py:79-92, compiled from the reference file's own source text."""
    path = os.path.join(mg.REF, "test", "test_sidd.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "expand2square")
    ns = {"torch": torch, "math": math}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["expand2square"]


def pooled(y: torch.Tensor, k: int = 16) -> torch.Tensor:
    return torch.nn.functional.avg_pool2d(y, k)


def build_ref(cfg, sd):
    m = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                    win_size=8, token_projection="linear", token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in).eval()
    m.load_state_dict(sd, strict=True)
    return m


@torch.no_grad()
def make_720p():
    e2s = reference_expand2square()
    img = spec.synth_input(1, 720, 1280, 9)
    xp, msk = e2s(img, factor=128)
    assert xp.shape == (1, 3, 1280, 1280)
    out = {"in_seed": 9, "x_sha256": hashlib.sha256(xp.numpy().tobytes()).hexdigest(), "mask_sum": float(msk.sum())}
    for ctor in (256, 128):
        cfg = spec.arch_config("Uformer_B", img_size=ctor)
        sd = spec.synth_state_dict(cfg, 1234)
        m = build_ref(cfg, sd)
        t0 = time.time()
        y = m(xp)
        frame = torch.masked_select(y, msk.bool()).reshape(1, 3, 720, 1280)            # test/test_sidd.py:108
        print(f"   reference Uformer-B(img_size={ctor}) on 1280x1280: {time.time() - t0:.1f} s, |y-x| max {(frame - img).abs().max():.4f}")
        tag = f"c{ctor}."
        for name, (r0, c0) in CROPS_720P.items():
            out[tag + "crop." + name] = frame[:, :, r0:r0 + 64, c0:c0 + 64].clone()
        out[tag + "pooled16"] = pooled(frame)
        out[tag + "sum"] = frame.sum((0, 2, 3)).double()
        out[tag + "abs_sum"] = frame.abs().sum((0, 2, 3)).double()
        out[tag + "max"] = frame.amax((0, 2, 3))
        out[tag + "min"] = frame.amin((0, 2, 3))
        out[tag + "sd_sha256"] = mg.sd_digest(sd)
        # pad region of the square output is not part of the contract (the scripts discard it) but pins expand2square's
        # offsets: keep the pooled map of the whole square too
        out[tag + "square_pooled32"] = pooled(y, 32)
    save("model_B_720p", **out)


def make_grad_B():
    sys.path.insert(0, mg.REF)
    import losses as ref_losses
    cfg = spec.arch_config("Uformer_B", img_size=256)
    sd = spec.synth_state_dict(cfg, 1234)
    m = build_ref(cfg, sd)
    xin = spec.synth_input(1, 256, 256, 1234).requires_grad_(True)
    target = spec.synth_input(1, 256, 256, 1235)
    t0 = time.time()
    y = m(xin)
    loss = ref_losses.CharbonnierLoss()(y, target)
    loss.backward()
    print(f"   reference fwd+bwd Uformer-B 256x256: {time.time() - t0:.1f} s, loss {float(loss):.6f}")
    names = [k for k, _ in m.named_parameters()]
    proj = np.zeros((len(names), 2), dtype=np.float64)
    norms = np.zeros((len(names), 2), dtype=np.float64)
    out = {}
    seen_kind = set()
    for i, (k, p_) in enumerate(m.named_parameters()):
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
            g2 = gr.reshape(gr.shape[0], -1)
            out["block64." + k] = g2[:64, :64].clone()
    save("grad_model_B_256", loss=loss.detach().double(), y_pooled16=pooled(y.detach()), y_crop=y.detach()[:, :, 96:160, 96:160],
         dx_crop=xin.grad[:, :, 96:160, 96:160], dx_pooled16=pooled(xin.grad), dx_abs_sum=float(xin.grad.abs().sum()),
         param_names=np.array(names), proj=proj, norms=norms, sd_sha256=mg.sd_digest(sd), **out)
    print(f"   {len(names)} parameters, {sum(1 for k in out if k.startswith('block64.'))} 64x64 blocks, "
          f"{sum(1 for k in out if k.startswith('full.'))} full tensors, {sum(1 for k in out if k.startswith('gather.'))} gathers")


if __name__ == "__main__":
    torch.set_num_threads(8)
    what = sys.argv[1:] or ["720p", "grad"]
    if "720p" in what:
        make_720p()
    if "grad" in what:
        make_grad_B()
