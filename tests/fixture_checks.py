"""Shared checkers for the round-2 reference fixtures (tests/golden/make_golden_r2.py), used by the CPU oracle tests and by the
GPU tests through the C ABI, so both are held to the reference's own numbers in exactly the same way."""
import numpy as np
import torch

from gradproj import gather_index, proj_vector

CROPS_720P = {"centre": (328, 608), "top_left": (0, 0), "bottom_right": (656, 1216), "top_mid": (0, 608), "left_mid": (328, 0)}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def check_720p(g, frame: torch.Tensor, ctor: int, tol: float, pooled_tol: float):
    """frame: (1,3,720,1280) restored frame (after the masked_select crop, before clamp).  Returns measured errors."""
    assert tuple(frame.shape) == (1, 3, 720, 1280) and torch.isfinite(frame).all()
    frame = frame.float().cpu()
    tag = f"c{ctor}."
    worst = 0.0
    for name, (r0, c0) in CROPS_720P.items():
        err = (frame[:, :, r0:r0 + 64, c0:c0 + 64] - _t(g[tag + "crop." + name])).abs().max().item()
        assert err <= tol, f"720p crop {name} (ctor {ctor}): {err:.3e} > {tol}"
        worst = max(worst, err)
    pe = (torch.nn.functional.avg_pool2d(frame, 16) - _t(g[tag + "pooled16"])).abs().max().item()
    assert pe <= pooled_tol, f"720p pooled map (ctor {ctor}): {pe:.3e} > {pooled_tol}"
    s = frame.sum((0, 2, 3)).double()
    ref_s, ref_abs = _t(g[tag + "sum"]), _t(g[tag + "abs_sum"])
    assert ((s - ref_s).abs() <= pooled_tol * 720 * 1280).all()        # mean error per pixel under the pooled tolerance
    assert ((frame.amax((0, 2, 3)) - _t(g[tag + "max"])).abs() <= tol).all()
    assert ((frame.amin((0, 2, 3)) - _t(g[tag + "min"])).abs() <= tol).all()
    return {"crop_max_abs_err": worst, "pooled16_max_abs_err": pe, "mean_abs_sum_rel": float(((frame.abs().sum((0, 2, 3)).double() - ref_abs).abs() / ref_abs).max())}


def check_grad_B(g, loss: float, y: torch.Tensor, dx: torch.Tensor, grads: dict, *, rtol: float, loss_tol: float, y_tol: float):
    """Every parameter gradient of Uformer-B 256x256 under the reference's Charbonnier loss: two signed random projections
    (tolerance rtol * ||g_ref||_2, the standard deviation a relative element error rtol gives them), a seeded 4096-element
    gather or the full tensor (rtol * max|g_ref|) and, for one parameter of every kind per stage, the top-left 64x64 block.
    Returns the worst relative deviations."""
    assert abs(loss - float(g["loss"])) <= loss_tol, (loss, float(g["loss"]))
    y, dx = y.float().cpu(), dx.float().cpu()
    assert (y[:, :, 96:160, 96:160] - _t(g["y_crop"])).abs().max().item() <= y_tol
    assert (torch.nn.functional.avg_pool2d(y, 16) - _t(g["y_pooled16"])).abs().max().item() <= y_tol
    dref = _t(g["dx_crop"])
    assert (dx[:, :, 96:160, 96:160] - dref).abs().max().item() <= rtol * dref.abs().max().item()
    pref = _t(g["dx_pooled16"])
    assert (torch.nn.functional.avg_pool2d(dx, 16) - pref).abs().max().item() <= rtol * max(pref.abs().max().item(), dref.abs().max().item() / 16)
    names = [str(n) for n in g["param_names"]]
    assert sorted(names) == sorted(k for k in grads), "parameter set differs from the reference's named_parameters()"
    return check_param_grads(g, grads, rtol, min_blocks=40)


def check_param_grads(g, grads: dict, rtol: float, min_blocks: int):
    """Every parameter gradient against the probes a fixture generator stored (tests/gradproj.py): two signed random projections
    (tolerance rtol * ||g_ref||_2), a seeded 4096-element gather or the full tensor (rtol * max|g_ref|) and 64x64 blocks of one
    parameter per kind and stage.  A gradient that is None counts as zeros (a block DropPath removed for every sample)."""
    names = [str(n) for n in g["param_names"]]
    assert sorted(names) == sorted(k for k in grads), "parameter set differs from the reference's named_parameters()"
    worst = {"proj": 0.0, "elem": 0.0, "block": 0.0}
    nfull = ngather = nblock = 0
    for i, n in enumerate(names):
        gr = grads[n]
        l2ref = float(g["norms"][i, 0])
        assert gr is not None or l2ref == 0.0, n
        if gr is None:
            continue
        gr = gr.detach().float().cpu()
        l2, mx = float(g["norms"][i, 0]), float(g["norms"][i, 1])
        for k in range(2):
            got = float((gr.double() * proj_vector(n, k, gr.shape).double()).sum())
            dev = abs(got - float(g["proj"][i, k])) / max(l2, 1e-30)
            assert dev <= rtol, f"{n}: projection {k} off by {dev:.3e} x ||g|| (tol {rtol})"
            worst["proj"] = max(worst["proj"], dev)
        if ("full." + n) in g:
            ref = _t(g["full." + n]); got = gr; nfull += 1
        else:
            ref = _t(g["gather." + n]); got = gr.reshape(-1)[gather_index(n, gr.numel())]; ngather += 1
        dev = (got - ref).abs().max().item() / max(mx, 1e-30)
        assert dev <= rtol, f"{n}: elements off by {dev:.3e} x max|g| (tol {rtol})"
        worst["elem"] = max(worst["elem"], dev)
        if ("block64." + n) in g:
            ref = _t(g["block64." + n]); nblock += 1
            dev = (gr.reshape(gr.shape[0], -1)[:64, :64] - ref).abs().max().item() / max(mx, 1e-30)
            assert dev <= rtol, f"{n}: 64x64 block off by {dev:.3e} x max|g|"
            worst["block"] = max(worst["block"], dev)
    assert nblock >= min_blocks
    return worst


def check_grad_T(g, loss: float, y: torch.Tensor, dx: torch.Tensor, grads: dict, *, rtol: float, loss_tol: float, y_tol: float):
    """tests/golden/grad_model_T_128.npz (Uformer_T = head_dim 16, train mode, recorded DropPath masks): loss, restored images, d loss /
    d input and every parameter gradient (check_param_grads)."""
    assert abs(loss - float(g["loss"])) <= loss_tol, (loss, float(g["loss"]))
    y, dx = y.float().cpu(), dx.float().cpu()
    ey = (y - _t(g["y"])).abs().max().item()
    assert ey <= y_tol, f"restored images off by {ey:.3e} (tol {y_tol})"
    dref = _t(g["dx"])
    ed = (dx - dref).abs().max().item() / dref.abs().max().item()
    assert ed <= rtol, f"d loss / d input off by {ed:.3e} x max (tol {rtol})"
    worst = check_param_grads(g, grads, rtol, min_blocks=20)
    worst.update(y=ey, dx=ed)
    return worst
