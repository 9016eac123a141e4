"""SURVEY section 8f rows through the C ABI on the GPU: training-step tail (Charbonnier, AdamW), metrics (PSNR, SSIM), the
arbitrary-resolution wrapper (expand2square / crop + clamp / tiled restore) and the device input pipeline (crop + rot/flip, MixUp),
each against the reference's formula restated in the oracle or against torch's own CPU implementation."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import uformer_oracle as O
from uformer_amd import spec

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 37, 53), (32, 3, 256, 256)])
def test_charbonnier_loss_and_gradient(shape):
    """loss and d loss / d restored of losses.py:41-52 in one pass vs autograd through the oracle's formula; bit-reproducible."""
    from uformer_amd import losses, ops
    y = torch.rand(shape, generator=g(1)) * 1.2 - 0.1
    t = torch.rand(shape, generator=g(2))
    yr = y.clone().requires_grad_(True)
    ref = O.charbonnier_loss(yr, t)
    ref.backward()
    loss, dy = ops.charbonnier(y.cuda(), t.cuda())
    loss2, dy2 = ops.charbonnier(y.cuda(), t.cuda())
    assert torch.equal(loss, loss2) and torch.equal(dy, dy2)
    assert abs(loss.item() - ref.item()) <= 2e-7 * max(1.0, abs(ref.item()))
    assert (dy.cpu() - yr.grad).abs().max().item() <= 1e-6 * yr.grad.abs().max().item()
    # the nn.Module form with autograd on both arguments (train/train_denoise.py:181)
    a, b = y.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    l3 = losses.CharbonnierLoss()(a, b)
    (l3 * 3.0).backward()
    assert torch.allclose(a.grad.cpu(), 3.0 * yr.grad, rtol=1e-5, atol=1e-12) and torch.allclose(b.grad.cpu(), -3.0 * yr.grad, rtol=1e-5, atol=1e-12)


def test_adamw_three_steps_vs_torch_cpu():
    """uf_adamw_step vs torch.optim.AdamW on the CPU with the reference's hyper-parameters (train/train_denoise.py:77), 3 steps,
    110 tensors of assorted sizes (scalars, unaligned lengths, > one 8192-element chunk, > 40 tensors = several launches)."""
    from uformer_amd import optim
    sizes = [(1,), (3,), (7, 5), (64,), (225, 2), (128, 32), (8193,), (4, 1, 3, 3), (1024, 256)] + [(17 + i, 3) for i in range(101)]
    ps_cpu = [torch.randn(s, generator=g(10 + i)).requires_grad_(True) for i, s in enumerate(sizes)]
    ps_gpu = [p.detach().clone().cuda().requires_grad_(True) for p in ps_cpu]
    kw = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    oc, og = torch.optim.AdamW(ps_cpu, **kw), optim.AdamW(ps_gpu, **kw)
    for step in range(3):
        for i, (pc, pg) in enumerate(zip(ps_cpu, ps_gpu)):
            gr = torch.randn(pc.shape, generator=g(1000 * step + i)) * (10.0 ** ((i % 5) - 3))
            pc.grad, pg.grad = gr.clone(), gr.clone().cuda()
        oc.step(); og.step()
    for pc, pg in zip(ps_cpu, ps_gpu):
        assert torch.allclose(pg.detach().cpu(), pc.detach(), rtol=1e-6, atol=1e-9), (pc.shape, (pg.detach().cpu() - pc.detach()).abs().max())
    for pc, pg in zip(ps_cpu, ps_gpu):
        sc, sg = oc.state[pc], og.state[pg]
        assert int(sc["step"]) == int(sg["step"]) == 3
        assert torch.allclose(sg["exp_avg"].cpu(), sc["exp_avg"], rtol=1e-6, atol=1e-12)
        assert torch.allclose(sg["exp_avg_sq"].cpu(), sc["exp_avg_sq"], rtol=1e-6, atol=1e-14)
    # grad_scale = 1/world folds the all-reduce average: same as stepping on g/4
    p1, p2 = torch.randn(5000, generator=g(5)).cuda().requires_grad_(True), None
    p2 = p1.detach().clone().requires_grad_(True)
    o1, o2 = optim.AdamW([p1], **kw), optim.AdamW([p2], **kw)
    gr = torch.randn(5000, generator=g(6)).cuda()
    p1.grad, p2.grad = gr.clone(), gr / 4
    o1.step(grad_scale=0.25); o2.step()
    assert torch.allclose(p1, p2, rtol=1e-6, atol=1e-9)


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """{'epoch','state_dict','optimizer'} written by us loads into torch.optim.AdamW / a fresh model, and a dict written the
    reference's way (torch.optim.AdamW state, 'module.' prefix) resumes our optimizer (train/train_denoise.py:101-119,207-235)."""
    from uformer_amd import checkpoint, model, optim
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 3)
    m = model.Uformer(img_size=128, embed_dim=32, depths=list(cfg.depths), modulator=True, compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    opt = optim.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    for p in m.parameters():
        p.grad = torch.randn_like(p) * 1e-2
    opt.step()
    path = str(tmp_path / "model_latest.pth")
    checkpoint.save_training_state(path, 7, m, opt, data_parallel_prefix=True)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and all(k.startswith("module.") for k in ck["state_dict"])
    assert checkpoint.load_start_epoch(path) == 7
    m2 = model.Uformer(img_size=128, embed_dim=32, depths=list(cfg.depths), modulator=True, compute_dtype=torch.float32)
    checkpoint.load_checkpoint(m2, path)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    ref_opt = torch.optim.AdamW(m2.parameters(), lr=1.0)            # the reference's optimizer class resumes from our file
    assert checkpoint.load_optim(ref_opt, path) == 2e-4
    m2 = m2.cuda()
    opt2 = optim.AdamW(m2.parameters(), lr=1.0)
    buf = io.BytesIO()
    torch.save({"epoch": 8, "state_dict": m2.state_dict(), "optimizer": ref_opt.state_dict()}, buf)    # a reference-written file
    p2 = str(tmp_path / "ref.pth")
    open(p2, "wb").write(buf.getvalue())
    assert checkpoint.load_optim(opt2, p2) == 2e-4
    for p in m.parameters():
        p.grad = torch.full_like(p, 1e-3)
    for p in m2.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step(); opt2.step()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)


def test_dynamic_loss_scaler_matches_torch_gradscaler_semantics():
    """uformer_amd.optim.GradScaler + AdamW (all decisions on the device: uf_grad_scaler_check / uf_adamw_step_scaled / uf_grad_scaler_update) against
    the protocol of torch.cuda.amp.GradScaler the reference trains under (train/train_denoise.py:180-184), restated on the CPU: unscale, skip the
    optimizer step and halve the scale when a gradient is inf / nan, double the scale after ``growth_interval`` clean steps, and -- because a skipped
    step never reaches the optimizer -- Adam's bias corrections count only the steps actually taken."""
    from uformer_amd import optim as uo
    torch.manual_seed(11)
    shapes = [(33, 7), (5,), (4, 3, 3, 3), (70000,)]
    params = [torch.randn(*sh) for sh in shapes]
    ref_p = [torch.nn.Parameter(p.clone()) for p in params]
    got_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    ref_opt = torch.optim.AdamW(ref_p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    opt = uo.AdamW(got_p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    scaler = uo.GradScaler(init_scale=1024.0, growth_interval=3)
    ref_scale, tracker, taken = 1024.0, 0, 0
    overflow_at = {2: float("inf"), 5: float("nan")}                     # steps whose gradients overflow
    for step in range(9):
        grads = [torch.randn(*sh, generator=torch.Generator().manual_seed(100 * step + i)) for i, sh in enumerate(shapes)]
        scaled = [g * ref_scale for g in grads]                           # what backward() of the scaled loss leaves in .grad
        if step in overflow_at:
            scaled[step % len(shapes)].view(-1)[3] = overflow_at[step]
        for p_, g in zip(got_p, scaled):
            p_.grad = g.clone().cuda()
        assert abs(scaler.get_scale() - ref_scale) == 0.0
        scaler.step(opt)
        scaler.update()
        # reference protocol on the CPU
        if any(not torch.isfinite(g).all() for g in scaled):
            ref_scale *= 0.5; tracker = 0
        else:
            for p_, g in zip(ref_p, scaled):
                p_.grad = g / ref_scale
            ref_opt.step(); taken += 1; tracker += 1
            if tracker == 3:
                ref_scale *= 2.0; tracker = 0
        for a, b in zip(got_p, ref_p):
            assert torch.allclose(a.detach().cpu(), b.detach(), rtol=2e-6, atol=1e-8), (step, (a.detach().cpu() - b.detach()).abs().max().item())
    assert scaler.steps_taken() == taken == 7 and abs(scaler.get_scale() - ref_scale) == 0.0
    scaler.sync_steps(opt)
    assert all(int(st["step"]) == 7 for st in opt.state.values())
    sd = scaler.state_dict()
    s2 = uo.GradScaler(); s2.load_state_dict(sd)
    assert s2.get_scale() == scaler.get_scale() and s2.steps_taken() == 7


def test_scaled_step_resumes_bias_correction_from_loaded_optimizer_state():
    """a reference checkpoint carries the optimizer (per-parameter step / exp_avg / exp_avg_sq) but no scaler
    (train/train_denoise.py:207-235).  Loading it at step N and training on under a fresh GradScaler must continue the bias corrections at
    N + 1, and the state_dict written afterwards must carry N + k -- compared with torch.optim.AdamW doing the same on the CPU."""
    from uformer_amd import optim as uo
    shapes = [(17, 5), (9,), (3000,)]
    gen = torch.Generator().manual_seed(77)
    params = [torch.randn(*sh, generator=gen) for sh in shapes]
    ref_p = [torch.nn.Parameter(p.clone()) for p in params]
    ref_opt = torch.optim.AdamW(ref_p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    grads = lambda k: [torch.randn(*sh, generator=torch.Generator().manual_seed(1000 + 10 * k + i)) for i, sh in enumerate(shapes)]   # noqa: E731
    for k in range(5):                                                   # N = 5 steps "before the checkpoint"
        for p_, g_ in zip(ref_p, grads(k)):
            p_.grad = g_
        ref_opt.step()
    ckpt = ref_opt.state_dict()
    got_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    opt = uo.AdamW(got_p, lr=9.9)
    opt.load_state_dict(ckpt)                                            # moves the moments to the parameters' device, as torch's does
    scaler = uo.GradScaler(init_scale=256.0)
    for k in range(5, 8):                                                # three scaled steps after the resume
        for p_, g_ in zip(got_p, grads(k)):
            p_.grad = (g_ * 256.0).cuda()
        scaler.step(opt); scaler.update()
        for p_, g_ in zip(ref_p, grads(k)):
            p_.grad = g_
        ref_opt.step()
        for a, b in zip(got_p, ref_p):
            assert torch.allclose(a.detach().cpu(), b.detach(), rtol=2e-6, atol=1e-8), (k, (a.detach().cpu() - b.detach()).abs().max().item())
    assert scaler.steps_taken() == 8
    sd = opt.state_dict()                                                # syncs the device-side count into the per-parameter entries by itself
    assert all(int(st["step"]) == 8 for st in sd["state"].values())
    for p_, g_ in zip(got_p, grads(8)):                                  # and a PLAIN step afterwards continues at 9
        p_.grad = g_.cuda()
    opt.step()
    for p_, g_ in zip(ref_p, grads(8)):
        p_.grad = g_
    ref_opt.step()
    for a, b in zip(got_p, ref_p):
        assert torch.allclose(a.detach().cpu(), b.detach(), rtol=2e-6, atol=1e-8)
    assert all(int(st["step"]) == 9 for st in opt.state.values())


def test_batch_psnr_and_ssim_vs_reference_formulas():
    from uformer_amd import metrics
    a = torch.rand(3, 3, 72, 200, generator=g(1)) * 1.3 - 0.15             # values outside [0,1]: the clamp matters
    b = (a + 0.05 * torch.randn(a.shape, generator=g(2))).clamp(-0.2, 1.2)
    ref = O.batch_psnr(a, b, average=False)
    got = metrics.batch_PSNR(a.cuda(), b.cuda(), average=False).item()
    assert abs(got - ref) <= 1e-4 * abs(ref)
    assert abs(metrics.myPSNR(a[0].cuda(), b[0].cuda()).item() - O.psnr(a[0], b[0])) <= 1e-4
    s = metrics.batch_SSIM(a.cuda(), b.cuda(), average=False).cpu()
    for i in range(3):
        assert abs(s[i].item() - O.ssim(a[i], b[i])) <= 1e-5, (i, s[i].item(), O.ssim(a[i], b[i]))


def test_batch_ssim_vs_reference_calculate_ssim_fixture():
    """uf_batch_ssim against outputs of the reference's OWN calculate_ssim (utils/caculate_psnr_ssim.py:35-81, ast-compiled with a cv2 shim by
    tests/golden/make_golden_tail.py: tail_ssim.npz, pinned_by = "reference"), including the pair whose out-of-range values wrap."""
    import numpy as np
    from uformer_amd import metrics
    gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tail_ssim.npz"), allow_pickle=True)
    assert str(gz["pinned_by"]) == "reference"
    for ka, kb, kv in (("a", "b", "ssim"), ("a_wrap", "b_wrap", "ssim_wrap")):
        a, b = torch.from_numpy(gz[ka]), torch.from_numpy(gz[kb])
        s = metrics.batch_SSIM(a.cuda(), b.cuda(), average=False).cpu()
        for i in range(a.shape[0]):
            assert abs(s[i].item() - float(gz[kv][i])) <= 1e-5, (kv, i, s[i].item(), float(gz[kv][i]))


def test_expand2square_and_crop_clamp_kernels():
    from uformer_amd import ops
    for (h, w) in ((720, 1280), (200, 136), (128, 128), (130, 3)):
        img = spec.synth_input(2, h, w, 5) * 1.4 - 0.2
        ref_c, ref_m = O.expand2square(img[:1], 128.0)
        c, m = ops.expand2square(img.cuda(), 128.0)
        assert torch.equal(c[:1].cpu(), ref_c) and torch.equal(m[:1].cpu(), ref_m)
        ref_c1, _ = O.expand2square(img[1:], 128.0)
        assert torch.equal(c[1:].cpu(), ref_c1)
        back = ops.crop_clamp(c, h, w, clamp=True).cpu()
        exp = torch.clamp(torch.masked_select(ref_c, ref_m.bool()).reshape(1, 3, h, w), 0, 1)
        assert torch.equal(back[:1], exp)
        assert torch.equal(ops.crop_clamp(c, h, w, clamp=False).cpu(), img)


def test_crop_augment_and_mixup_vs_reference_transforms():
    from uformer_amd import data, ops
    N, H, W, ps = 5, 96, 80, 48
    frames_u8 = torch.randint(0, 256, (N, H, W, 3), generator=g(3), dtype=torch.uint8)       # as cv2 decodes: HWC uint8
    frames_f = (frames_u8.float() / 255.0).permute(0, 3, 1, 2).contiguous()                 # load_img + permute(2,0,1)
    meta = torch.tensor([[i % N, (7 * i) % (H - ps), (11 * i) % (W - ps), i % 8] for i in range(16)], dtype=torch.int32)
    ref = torch.stack([O.crop_augment(frames_f[i], r, c, ps, k) for i, r, c, k in meta.tolist()])
    got_u8 = ops.crop_augment(frames_u8.cuda(), meta.cuda(), ps, hwc=True).cpu()
    got_f = ops.crop_augment(frames_f.cuda(), meta.cuda(), ps, hwc=False).cpu()
    assert torch.equal(got_u8, ref) and torch.equal(got_f, ref)
    loader = data.GpuPatchLoader(frames_u8.cuda(), frames_u8.flip(0).cuda(), ps)
    cl, no = loader.batch(8)
    assert cl.shape == no.shape == (8, 3, ps, ps) and cl.min() >= 0 and cl.max() <= 1
    lam = torch.rand(16, generator=g(4))
    perm = torch.randperm(16, generator=g(5))
    assert torch.allclose(ops.mixup(ref.cuda(), lam.cuda(), perm.cuda().int()).cpu(), O.mixup(ref, lam, perm), rtol=0, atol=1e-7)
    a, b = data.MixUp_AUG().aug(cl, no)
    assert a.shape == cl.shape and torch.isfinite(a).all() and torch.isfinite(b).all()


def test_restore_tiled_matches_single_tile_and_tracks_full_frame():
    """restore_tiled == restore when the image fits one tile; on a 384x640 frame cut into overlapping 384-tiles it stays close
    to the full-frame (padded to 640x640) result -- it is an approximation by construction, the number is reported."""
    from uformer_amd import infer, model
    cfg = spec.arch_config("tiny32", img_size=128)
    sd = spec.synth_state_dict(cfg, 5)
    m = model.Uformer(img_size=128, embed_dim=32, depths=list(cfg.depths), modulator=True, compute_dtype=torch.float32).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    img = spec.synth_input(1, 200, 136, 11).cuda()
    assert torch.equal(infer.restore_tiled(m, img, tile=256), infer.restore(m, img))
    img = spec.synth_input(1, 384, 640, 12).cuda()
    full = infer.restore(m, img)
    tiled = infer.restore_tiled(m, img, tile=384, min_overlap=128)
    assert tiled.shape == full.shape and torch.isfinite(tiled).all()
    ps = O.psnr(tiled.cpu(), full.cpu())
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"psnr_tiled_vs_full_db": ps}, open("gpurun_out/parity_tiled.json", "w"))
    assert ps >= 30.0, ps


def _sink_on_gpu_worker(rank, world, port, q, algorithm, payload):
    """two ranks SHARE cuda:0 and meet over gloo (what a 1-GPU box allows): buckets of device tensors, the exchange chained on the sink's side stream
    behind the kernels that wrote them, AdamW's grad_scale path afterwards -- the stream plumbing of uformer_amd.dist on real hardware"""
    import sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from uformer_amd import dist as ud
    ud.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(7)
    shapes = [(512, 2048), (2048,), (300, 300), (64, 512), (1000,), (257, 129)]
    params = [(f"p{i}", torch.nn.Parameter(torch.zeros(s, device=dev))) for i, s in enumerate(shapes)]
    sink = ud.OverlappedGradientAllReduce(params, bucket_bytes=1 << 20, algorithm=algorithm, payload=getattr(torch, payload))
    worst = 0.0
    for step in range(3):
        local = {n: torch.randn(p.shape, generator=g) * (1 + rank) for n, p in params}           # both ranks draw the same stream: rank r holds (1 + r) x
        sink.begin_step()
        for n, p in reversed(params):
            t = local[n].to(dev)
            big = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)       # work in flight on the compute stream when the bucket is launched
            sink.deliver({n: t * 1.0})
            del big
        sink.finish()
        for n, p in params:
            want = local[n] * (sum(1 + r for r in range(world)) / (1 + rank))                     # sum over ranks of (1 + r) x
            got = p.grad.detach().float().cpu()
            worst = max(worst, float((got - want).abs().max() / want.abs().max()))
    q.put((rank, worst))
    dist.destroy_process_group()


@pytest.mark.parametrize("algorithm,payload,tol", [("ring", "float32", 1e-6), ("direct", "float32", 1e-6), ("direct", "bfloat16", 1.2e-2)])
def test_gradient_sink_on_device_tensors_two_ranks_one_gpu(algorithm, payload, tol):
    """uformer_amd.dist.OverlappedGradientAllReduce with DEVICE buckets (gloo, two ranks sharing the GPU): the all-reduce and the direct exchange (round 6: its
    side stream, the staging buffers, the join in finish()) give the sum over ranks while the compute stream is busy, three steps in a row."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sink_on_gpu_worker, args=(r, 2, port, q, algorithm, payload)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(w < tol for _, w in res), res
