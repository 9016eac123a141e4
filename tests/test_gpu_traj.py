"""K-step training trajectory through the nn.Module boundary on the native kernels against the REFERENCE's own loop (VERDICT r04 "next" 7):
fixture tests/golden/traj_tiny32_8steps.npz = reference Uformer (train mode) + CharbonnierLoss + torch.optim.AdamW over 8 steps with recorded
DropPath masks (train/train_denoise.py:175-184).  A single-step gradient test cannot see what this one does: a packed-weight cache that is not
refreshed after the in-place optimizer step, a bias-correction count that is off, scaler state that leaks between steps, a DropPath stream
that is consumed in the wrong order."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gradproj  # noqa: E402

pytestmark = pytest.mark.gpu

# loss-curve tolerance (relative, every step) and weight-change tolerance (projections, relative to delta_max * sqrt(numel)).  VERDICT r04 asked for
# 1e-5 (f32) and 2 % (2-byte types) on the loss curve; the gates are set at 2-4 x what the kernels measure on MI355X (profiles/r05_parity_traj_*.json:
# loss f32 8.8e-8 / f16 5.0e-5 / bf16 3.3e-4; weight-change projections 5.6e-4 / 9.3e-2 / 3.9e-1 -- Adam normalises every gradient element to O(lr), so
# operand rounding shows in the small-gradient elements of a tensor; final weights f32 1.3e-6)
LOSS_RTOL = {torch.float32: 1e-5, torch.float16: 2e-4, torch.bfloat16: 1e-3}
DELTA_TOL = {torch.float32: 2e-3, torch.float16: 0.2, torch.bfloat16: 0.6}
# final weights, relative to the largest weight of the tensor (the projection gates of the 2-byte types are loose -- Adam turns every gradient element
# into a step of O(lr), so a rounding-level sign flip of a tiny gradient moves a projection by a whole step -- while the WEIGHTS after 8 steps differ from the
# reference's by at most 8 steps x lr relative to weights of O(1)): measured f32 1.3e-6, f16 7.2e-3, bf16 8.1e-3 (a small-magnitude bias whose elements took
# opposite steps: 8 x 2e-4 against a largest weight of 0.2) -- the 2-byte gate is twice that saturation level; the step-count assertion below is the sharp one
FINAL_RTOL = {torch.float32: 1e-5, torch.float16: 1.6e-2, torch.bfloat16: 1.6e-2}
TAG = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_eight_step_trajectory_vs_reference_loop(golden, dtype):
    from uformer_amd import losses as ul
    from uformer_amd import model, spec
    from uformer_amd import optim as uo
    g = golden("traj_tiny32_8steps")
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 1234)
    m = model.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                      dd_in=cfg.dd_in, drop_path_rate=float(g["drop_path_rate"]), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    assert np.allclose(m.drop_path_rates(), g["drop_rates"], atol=1e-6)
    opt = uo.AdamW(m.parameters(), lr=float(g["lr"]), betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)     # train/train_denoise.py:77
    crit = ul.CharbonnierLoss()
    scaler = uo.GradScaler() if dtype == torch.float16 else None          # the reference's fp16 protocol (train/train_denoise.py:180-184)
    xs = [spec.synth_input(2, 128, 128, 9000 + i).cuda() for i in range(2)]
    ts = [spec.synth_input(2, 128, 128, 9100 + i).cuda() for i in range(2)]
    losses = []
    for k in range(int(g["steps"])):
        m._drop_scales_override = t(g["masks"][k]).cuda()
        opt.zero_grad(set_to_none=True)
        restored = m(xs[k % 2])
        loss = crit(restored, ts[k % 2])
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
        else:
            loss.backward()
            opt.step()
        losses.append(float(loss.detach()))
    if scaler is not None:
        assert scaler.steps_taken() == int(g["steps"])                      # no overflow at the initial scale on this workload
    loss_err = max(abs(a - float(b)) / float(b) for a, b in zip(losses, g["losses"]))
    y_err = (restored.detach().float().cpu() - t(g["y_last"])).abs().max().item()
    worst = (0.0, "")
    params = dict(m.named_parameters())
    for i, n in enumerate(str(n_) for n_ in g["param_names"]):
        d = params[n].detach().float().cpu() - sd[n]
        scale = float(g["delta_max"][i]) * d.numel() ** 0.5
        for j in range(2):
            pr = float((d.double() * gradproj.proj_vector(n, j, d.shape).double()).sum())
            worst = max(worst, (abs(pr - float(g["delta_proj"][i][j])) / max(scale, 1e-30), n))
    full = max((params[k_[6:]].detach().float().cpu() - t(g[k_])).abs().max().item() / max(float(np.abs(g[k_]).max()), 1e-30)
               for k_ in g if k_.startswith("final."))
    rec = {"dtype": TAG[dtype], "losses": losses, "reference_losses": [float(v) for v in g["losses"]], "max_rel_loss_err": loss_err,
           "y_last_max_abs_err": y_err, "worst_delta_projection_err": worst[0], "worst_delta_projection_param": worst[1], "final_weights_max_rel_err": full}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_traj_{TAG[dtype]}.json", "w") as f:
        json.dump(rec, f, indent=1)
    assert loss_err < LOSS_RTOL[dtype], rec
    assert worst[0] < DELTA_TOL[dtype], rec
    assert full < FINAL_RTOL[dtype], rec                                    # final weights (f32: within 1e-5 relative, VERDICT r04 "next" 7)
    # the optimizer counted every step once (a wrong bias-correction count or a step the scaler skipped silently would pass the loss gate of the 2-byte types)
    steps_seen = {int(st["step"]) for st in opt.state_dict()["state"].values()}
    assert steps_seen == {int(g["steps"])}, steps_seen
