#!/usr/bin/env python3
"""Round-3 golden fixtures for the steps EITHER SIDE of the hot path, generated from the reference's
own code so that the oracle helpers they pin (``O.psnr`` / ``O.batch_psnr`` / ``O.augment`` / ``O.crop_augment`` / ``O.mixup`` /
``O.ssim``) and ``uformer_amd.checkpoint`` are no longer "parity unpinned" (VERDICT r02 "missing" 4):

  * ``tail_psnr``     -- ``myPSNR`` / ``batch_PSNR`` (utils/image_utils.py:40-51), compiled from the reference file's own source text
    with ``ast`` (the module itself imports ``cv2``, which is not installed);
  * ``tail_augment``  -- all 8 ``Augment_RGB_torch.transform<k>`` and ``MixUp_AUG.aug`` (utils/dataset_utils.py:5-49), the classes
    compiled the same way.  ``aug`` ends its lambda draw with ``.cuda()``; there is no GPU in the build container, so
    ``torch.Tensor.cuda`` is the identity while it runs (the draw itself is CPU torch in the reference too);
  * ``tail_checkpoint`` -- a checkpoint written by ``uformer_amd.checkpoint.save_training_state`` (plain and ``module.``-prefixed)
    is read by the REFERENCE's ``load_checkpoint`` into the REFERENCE's ``Uformer`` and by ``load_optim`` / ``load_start_epoch``
    (utils/model_utils.py:23-54, loaded by file path: it imports only torch / os / collections); the fixture stores the digest of
    what the reference holds afterwards, the test rebuilds the same file and compares;
  * ``tail_ssim``     -- round 5: the reference's own ``calculate_ssim`` / ``_ssim`` (utils/caculate_psnr_ssim.py:35-81), compiled with ``ast``
    and run with a 2-function ``cv2`` shim (``getGaussianKernel``, ``filter2D`` by OpenCV's documented semantics; cv2 is not installed
    here) -- ``pinned_by = "reference"``; the independent separable restatement of round 3 is kept as a cross-check (1e-9).

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tail.py
"""
import ast
import hashlib
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (timm shim + the reference's model.py as mg.ref)

ref, spec, save = mg.ref, mg.spec, mg.save


def compile_from(path, names, ns):
    """exec the top-level functions / classes ``names`` of the reference file ``path`` (its own source text) into ``ns``"""
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def make_psnr():
    ns = compile_from(os.path.join(mg.REF, "utils", "image_utils.py"), ["myPSNR", "batch_PSNR"], {"torch": torch})
    g = torch.Generator().manual_seed(301)
    a = torch.rand(5, 3, 40, 56, generator=g) * 1.3 - 0.15          # values outside [0,1] exercise the clamps
    b = (a + 0.05 * torch.randn(5, 3, 40, 56, generator=g)).clamp(-0.2, 1.2)
    per = torch.stack([ns["myPSNR"](x, y) for x, y in zip(a, b)])
    save("tail_psnr", a=a, b=b, per_image=per, avg=ns["batch_PSNR"](a, b, True), total=ns["batch_PSNR"](a, b, False),
         whole=ns["myPSNR"](a, b), source="utils/image_utils.py:40-51 (ast-compiled myPSNR, batch_PSNR)")


def make_augment():
    ns = compile_from(os.path.join(mg.REF, "utils", "dataset_utils.py"), ["Augment_RGB_torch", "MixUp_AUG"], {"torch": torch, "os": os})
    aug = ns["Augment_RGB_torch"]()
    g = torch.Generator().manual_seed(302)
    x = torch.rand(3, 6, 6, generator=g)                              # square patch (the data loader crops ps x ps, dataset_denoise.py:54-70)
    frame = torch.rand(3, 20, 28, generator=g)
    out = {"x": x, "frame": frame, "crop_r": 5, "crop_c": 9, "crop_ps": 8}
    names = [m for m in dir(aug) if callable(getattr(aug, m)) and m.startswith("transform")]    # dataset_denoise.py:14 builds the list this way
    assert names == [f"transform{k}" for k in range(8)], names
    for k in range(8):
        out[f"t{k}"] = getattr(aug, f"transform{k}")(x)
        out[f"crop_t{k}"] = getattr(aug, f"transform{k}")(frame[:, 5:13, 9:17])
    # MixUp: seed -> randperm -> Beta(1.2, 1.2).rsample, exactly the draws aug() makes; replayed here to record perm and lam
    bs = 6
    gt, noisy = torch.rand(bs, 3, 8, 8, generator=g), torch.rand(bs, 3, 8, 8, generator=g)
    mix = ns["MixUp_AUG"]()
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                   # no GPU here; the reference draws lam on the CPU and moves it
    try:
        torch.manual_seed(4242)
        gt2, noisy2 = mix.aug(gt, noisy)
    finally:
        torch.Tensor.cuda = keep
    torch.manual_seed(4242)
    perm = torch.randperm(bs)
    lam = mix.dist.rsample((bs, 1)).view(-1)
    out.update(mix_gt=gt, mix_noisy=noisy, mix_perm=perm, mix_lam=lam, mix_gt_out=gt2, mix_noisy_out=noisy2)
    save("tail_augment", source="utils/dataset_utils.py:5-49 (ast-compiled Augment_RGB_torch, MixUp_AUG)", **out)


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k].detach().cpu().contiguous()
        h.update(k.encode()); h.update(str(v.dtype).encode()); h.update(str(tuple(v.shape)).encode()); h.update(v.numpy().tobytes())
    return h.hexdigest()


def optim_digest(osd):
    h = hashlib.sha256()
    for gidx, gp in enumerate(osd["param_groups"]):
        h.update(repr(sorted((k, v) for k, v in gp.items() if k in ("lr", "betas", "eps", "weight_decay", "params"))).encode())
    for k in sorted(osd["state"]):
        for name in ("step", "exp_avg", "exp_avg_sq"):
            v = osd["state"][k][name]
            v = v.detach().cpu().float().contiguous() if torch.is_tensor(v) else torch.tensor(float(v))
            h.update(f"{k}.{name}".encode()); h.update(v.numpy().tobytes())
    return h.hexdigest()


def write_reference_style_checkpoint(path, prefix):
    """The file both sides must agree on, built ONLY with our code: tiny32 model with synthetic weights, one AdamW step on a fixed
    synthetic gradient (so exp_avg / exp_avg_sq / step are non-trivial), saved by uformer_amd.checkpoint.save_training_state."""
    from uformer_amd import checkpoint as ck
    from uformer_amd import model as um
    cfg = spec.arch_config("tiny32", img_size=128)
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                   dd_in=cfg.dd_in, compute_dtype=torch.float32)
    m.load_state_dict(spec.synth_state_dict(cfg, 77), strict=True)
    opt = torch.optim.AdamW(m.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)     # train/train_denoise.py:77 form
    gen = torch.Generator().manual_seed(78)
    for p in m.parameters():
        p.grad = 1e-3 * torch.randn(p.shape, generator=gen)
    opt.step()
    ck.save_training_state(path, 17, m, opt, data_parallel_prefix=prefix)
    return cfg, m, opt


def make_checkpoint():
    spec_mu = importlib.util.spec_from_file_location("ref_model_utils", os.path.join(mg.REF, "utils", "model_utils.py"))
    mu = importlib.util.module_from_spec(spec_mu)
    spec_mu.loader.exec_module(mu)
    out = {"source": "utils/model_utils.py:23-54 (load_checkpoint, load_start_epoch, load_optim; module loaded by file path)"}
    for prefix in (False, True):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "model_latest.pth")
            cfg, ours, opt = write_reference_style_checkpoint(path, prefix)
            rm = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), win_size=8,
                             token_projection="linear", token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in)
            mu.load_checkpoint(rm, path)                                   # the reference's own loader (strips module.)
            ropt = torch.optim.AdamW(rm.parameters(), lr=9.9, betas=(0.5, 0.5), eps=1.0, weight_decay=0.5)
            lr = mu.load_optim(ropt, path)
            epoch = mu.load_start_epoch(path)
            tag = "dp." if prefix else "plain."
            out[tag + "state_digest"] = sd_digest(rm.state_dict())
            out[tag + "optim_digest"] = optim_digest(ropt.state_dict())
            out[tag + "lr"] = float(lr)
            out[tag + "epoch"] = int(epoch)
            out[tag + "n_keys"] = len(rm.state_dict())
            assert sd_digest(rm.state_dict()) == sd_digest(ours.state_dict())
            # and the other direction: the reference's save_checkpoint file is read by our loader
            mu.save_checkpoint(d, {"epoch": 23, "state_dict": rm.state_dict(), "optimizer": ropt.state_dict()}, "sess")
            from uformer_amd import checkpoint as ck
            from uformer_amd import model as um
            back = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                              dd_in=cfg.dd_in, compute_dtype=torch.float32)
            ck.load_checkpoint(back, os.path.join(d, "model_epoch_23_sess.pth"))
            assert sd_digest(back.state_dict()) == sd_digest(rm.state_dict())
            out[tag + "roundtrip_from_reference_save"] = 1
    save("tail_checkpoint", **out)


def ssim_restated(img1_chw, img2_chw):
    """utils/caculate_psnr_ssim.py:35-81 line by line, with the two cv2 calls restated: getGaussianKernel(11, 1.5) = normalised
    exp(-(i-5)^2 / (2 sigma^2)); filter2D(img, -1, outer(k, k))[5:-5, 5:-5] = the 'valid' correlation, done separably in float64."""
    def to_u8(x):
        return (x.clamp(0, 1).numpy().astype(np.float32) * 255.0).round().astype(np.uint8)
    a = to_u8(img1_chw).transpose(1, 2, 0).astype(np.float64)          # reorder_image -> HWC
    b = to_u8(img2_chw).transpose(1, 2, 0).astype(np.float64)
    k = np.array([np.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)])
    k = k / k.sum()

    def valid(z):                                                       # rows then columns, valid region only
        H, W = z.shape
        tmp = np.zeros((H - 10, W))
        for i in range(11):
            tmp += k[i] * z[i:i + H - 10, :]
        o = np.zeros((H - 10, W - 10))
        for j in range(11):
            o += k[j] * tmp[:, j:j + W - 10]
        return o
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(a.shape[2]):
        x, y = a[..., c], b[..., c]
        mu1, mu2 = valid(x), valid(y)
        s1, s2, s12 = valid(x * x) - mu1 * mu1, valid(y * y) - mu2 * mu2, valid(x * y) - mu1 * mu2
        vals.append((((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean())
    return float(np.array(vals).mean())


class Cv2Shim:
    """The two cv2 functions utils/caculate_psnr_ssim.py:35-52 calls, by OpenCV's documented semantics (cv2 itself is not installed in the
    build container; this is the same kind of shim as the 3-symbol ``timm`` one that lets model.py import):
      * ``getGaussianKernel(ksize, sigma)``: (ksize, 1) float64, G_i = alpha * exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)), sum = 1
        (ksize 11 / sigma 1.5 is outside OpenCV's fixed small-kernel tables);
      * ``filter2D(src, -1, kernel)``: CORRELATION (no kernel flip), anchor at the kernel centre, same depth as the source,
        default border BORDER_REFLECT_101 = scipy's ``mode="mirror"`` (the reference crops [5:-5, 5:-5], so the border never shows)."""

    @staticmethod
    def getGaussianKernel(ksize, sigma):
        assert ksize % 2 == 1 and sigma > 0
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)

    @staticmethod
    def filter2D(src, ddepth, kernel):
        from scipy.ndimage import correlate
        assert ddepth == -1
        return correlate(src, kernel, mode="mirror")


def make_ssim():
    """tail_ssim: the reference's OWN ``calculate_ssim`` / ``_ssim`` / ``reorder_image`` (utils/caculate_psnr_ssim.py:35-81, :155-162), compiled from
    the reference file's source text and run with the cv2 shim above.  The second restatement stays in the fixture as a cross-check."""
    try:
        import cv2 as real_cv2  # noqa: F401
        have_cv2 = True
    except ImportError:
        real_cv2, have_cv2 = None, False
    ns = compile_from(os.path.join(mg.REF, "utils", "caculate_psnr_ssim.py"), ["_ssim", "calculate_ssim", "reorder_image"],
                      {"np": np, "cv2": real_cv2 if have_cv2 else Cv2Shim, "torch": torch})
    g = torch.Generator().manual_seed(303)
    a = torch.rand(3, 3, 37, 45, generator=g)
    b = (a + 0.08 * torch.randn(3, 3, 37, 45, generator=g)).clamp(0, 1)
    # one pair with values outside [0, 1]: the reference's (img * 255.0).round().astype(np.uint8) wraps them (:59-62)
    a2 = torch.rand(1, 3, 24, 31, generator=g) * 1.2 - 0.1
    b2 = a2 + 0.05 * torch.randn(1, 3, 24, 31, generator=g)
    ref_vals = [float(ns["calculate_ssim"](x.numpy(), y.numpy(), crop_border=0, input_order="CHW")) for x, y in zip(a, b)]
    ref_vals2 = [float(ns["calculate_ssim"](x.numpy(), y.numpy(), crop_border=0, input_order="CHW")) for x, y in zip(a2, b2)]
    restated = [ssim_restated(x, y) for x, y in zip(a, b)]
    assert max(abs(u - v) for u, v in zip(ref_vals, restated)) < 1e-9, (ref_vals, restated)
    save("tail_ssim", a=a, b=b, ssim=np.array(ref_vals), a_wrap=a2, b_wrap=b2, ssim_wrap=np.array(ref_vals2), ssim_restated=np.array(restated),
         cv2_available=int(have_cv2), pinned_by="reference",
         source="utils/caculate_psnr_ssim.py:35-81, :155-162: the reference's own calculate_ssim / _ssim / reorder_image, ast-compiled from its source text; "
                + ("cv2 is the installed OpenCV" if have_cv2 else
                   "cv2.getGaussianKernel / cv2.filter2D come from a 2-function shim written from OpenCV's documented semantics (cv2 is not installed here)"))


if __name__ == "__main__":
    torch.set_num_threads(4)
    make_psnr()
    make_augment()
    make_checkpoint()
    make_ssim()
