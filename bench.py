#!/usr/bin/env python3
"""Headline benchmark: restored images/s, Uformer-B 256x256, bf16 inference, batch 16 per GPU
(BASELINE.json configs[1]).  One process per GPU; the batch is sharded across ranks with no
data-path collective (weak scaling: per-GPU work is fixed).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of the hot path (Uformer.forward through uf_uformer_fwd) over one batch of
synthetic images already resident in HBM.  Rank 0 prints ONE JSON line.  Besides the contract
fields it carries
  * ``modes``: throughput AND parity of every operand type -- bf16 (the headline: BASELINE configs[1] names it), f16 (IEEE
    half, the reference's own AMP type: same MFMA rate, meets the 1e-3 north-star tolerance) and f32 (exact-f32 MFMA), each
    against the oracle on the same image; and ``modes.train``: BASELINE configs[2], Uformer-B 256x256 batch 32 forward +
    backward + Charbonnier + AdamW on the native kernels, with its own ``roofline`` (dominant instrumented backward kernel)
    and ``cpu_baseline`` (the oracle under torch autograd + torch.optim.AdamW at batch 2 on this host);
  * ``roofline``: the dominant kernel, from HIP events on the launch stream (library instrumentation);
    ``traffic`` (HBM bytes per launch from PMC counters) only when profiles/r02_pmc_traffic.json was
    measured on exactly these kernel sources (stamp check), else null;
  * ``cpu_baseline``: the oracle (CPU restatement) on this host: thread count swept, B = 1 and B = 4,
    median of 3 -- plus, for reference, the reference's own model.py as timed in the build container
    (``reference_container``, from the newest profiles/r*_reference_cpu.json; /root/reference does not exist here);
  * ``--error-budget``: bf16-mode error by source (oracle/bf16_budget.py), one switch at a time.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks (same guide)
TORCH_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
MB_PER_IMAGE_B256 = 184.8       # compulsory bf16 activation bytes per image (blocks in+out once, samplers, skips, stem/head)
MB_WEIGHTS_B = 101.8            # bf16 weights, read once per batch


def kernel_source_sha() -> str:
    """Identity of the kernels a measurement belongs to: SHA-256 over the HIP sources and headers."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "uformer_amd", "csrc", "*")) + [os.path.join(REPO, "include", "uformer_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel_breakdown(model, x, steps, fn=None):
    """Per-kernel-class time over `steps` forwards (or calls of ``fn``), measured with HIP events on the launch stream
    (library-side instrumentation, uf_timing_enable)."""
    from uformer_amd import _lib
    lib = _lib.load()
    torch.cuda.synchronize()
    lib.uf_timing_enable(1)
    # one stream while the events are in: a kernel that shares the GPU with another stream's kernel is timed at its share of the chip (linear_wgrad
    # 245 us inside the two-stream training step, 105 us alone: profiles/r04_run11.txt / r04_run12.txt).  The forward already runs on one stream
    # under uf_timing_enable; this covers the weight-gradient side stream of the op-by-op backward (read per block: uformer_amd/train.py _Side)
    keep = os.environ.get("UF_BWD_STREAMS")
    os.environ["UF_BWD_STREAMS"] = "1"
    try:
        if fn is not None:
            for _ in range(steps):
                fn()
        else:
            with torch.no_grad():
                for _ in range(steps):
                    model(x)
        torch.cuda.synchronize()
    finally:
        if keep is None:
            os.environ.pop("UF_BWD_STREAMS", None)
        else:
            os.environ["UF_BWD_STREAMS"] = keep
    lib.uf_timing_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.uf_timing_report(buf, len(buf))
    rows = json.loads(buf.value.decode())
    for r in rows:
        r["ms_per_launch"] = r["ms"] / max(1, r["launches"])
    rows.sort(key=lambda r: -r["ms"])
    return rows


def cpu_baseline(arch, img, budget_s=28.0):
    """The oracle (CPU restatement of the reference forward, fp32, eval) timed on this host.  Thread counts {8, 16, 32,
    physical cores} are probed once each at B = 1; the best one is then timed at B = 1 (median of 3) and at B = 4 (median of
    up to 3, while the time budget lasts); the better of the two is the reported value."""
    from oracle import uformer_oracle as O
    from uformer_amd import spec
    cfg = spec.arch_config(arch, img_size=img)
    sd = spec.synth_state_dict(cfg, 1234)
    x1 = spec.synth_input(1, img, img, 1234)
    x4 = spec.synth_input(4, img, img, 1234)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    logical = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = logical
    physical = max(1, avail // 2)
    cands = sorted({t for t in (8, 16, 32, physical) if 1 <= t <= avail} or {avail})
    t_start = time.perf_counter()

    def run(x):
        t0 = time.perf_counter()
        y = O.uformer_forward(x, sd, **kw)
        return time.perf_counter() - t0, y

    probe = {}
    with torch.no_grad():
        torch.set_num_threads(cands[0])
        _, ref = run(x1)                                   # warm-up (allocator, oneDNN primitives), also the parity reference
        for t in cands:
            torch.set_num_threads(t)
            probe[t] = run(x1)[0]
        best_t = min(probe, key=probe.get)
        torch.set_num_threads(best_t)
        b1 = [run(x1)[0] for _ in range(3)]
        b4 = []
        while len(b4) < 3 and (not b4 or time.perf_counter() - t_start + statistics.median(b4) < budget_s):
            b4.append(run(x4)[0])
    v1, v4 = 1.0 / statistics.median(b1), 4.0 / statistics.median(b4)
    out = {"value": max(v1, v4), "unit": "images/s", "cores": best_t, "kind": "port",
           "sample": (f"oracle/uformer_oracle.py ({arch} {img}x{img}, fp32): thread probe {{{', '.join(f'{t}: {1 / s:.2f} img/s' for t, s in probe.items())}}}; "
                      f"at {best_t} threads B=1 median of 3 = {v1:.3f} img/s, B=4 median of {len(b4)} = {v4:.3f} img/s; "
                      f"{time.perf_counter() - t_start:.0f} s of CPU work on {avail} available logical cores"),
           "b1_images_per_s": v1, "b4_images_per_s": v4, "threads_probed": {str(t): 1.0 / s for t, s in probe.items()}}
    try:                                                   # the reference ITSELF, timed where /root/reference exists
        newest = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_reference_cpu.json"))[-1]      # r06_reference_cpu.json (round 2: r02_...)
        with open(os.path.join(REPO, "profiles", newest)) as f:
            out["reference_container"] = json.load(f)
    except (OSError, ValueError):
        pass
    return out, (x1, ref)


def vendor_baseline(arch, img, batch, dev, x1=None, ref=None):
    """Same-node calibration (VERDICT r04 "next" 5): the reference's op sequence on the vendor stack of THIS box -- PyTorch-ROCm eager
    (hipBLASLt / rocBLAS linears, MIOpen convolutions, ATen LayerNorm / GELU / softmax / permutes), oracle/vendor_forward.py -- at the
    headline batch, fp32 and under bf16 autocast, outside every timed region of the hand-written path.  Not the oracle and not a target:
    it answers "does the hand-written HIP path beat PyTorch-ROCm on this forward on this GPU"."""
    from oracle import vendor_forward as V
    from uformer_amd import spec
    cfg = spec.arch_config(arch, img_size=img)
    sd = {k: v.to(dev) for k, v in spec.synth_state_dict(cfg, 1234).items()}
    x = spec.synth_input(batch, img, img, 1234).to(dev)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    out = {"what": f"oracle/vendor_forward.py: the reference forward (model.py:1269-1305 op sequence) in PyTorch-ROCm eager on this GPU, {arch} {img}x{img} batch {batch}",
           "torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "unit": "images/s"}
    for name, ctx in (("fp32", None), ("bf16_autocast", torch.bfloat16), ("f16_autocast", torch.float16)):
        try:
            def run():
                with torch.no_grad():
                    if ctx is None:
                        return V.forward(x, sd, **kw)
                    with torch.autocast("cuda", dtype=ctx):
                        return V.forward(x, sd, **kw)
            for _ in range(2):
                y = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                y = run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            ent = {"images_per_s": batch / dt, "ms_per_step": 1e3 * dt, "steps": n}
            if x1 is not None and ref is not None:       # its own error against the CPU oracle on the parity image
                with torch.no_grad():
                    if ctx is None:
                        y1 = V.forward(x1.to(dev), sd, **kw)
                    else:
                        with torch.autocast("cuda", dtype=ctx):
                            y1 = V.forward(x1.to(dev), sd, **kw)
                ent["max_abs_err_vs_oracle"] = float((y1.float().cpu() - ref).abs().max())
            out[name] = ent
            del y
        except Exception as e:   # noqa: BLE001 -- a vendor-library failure must not take the bench line down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
    return out


def vendor_baseline_guarded(args, timeout_s=240):
    """vendor_baseline() in a child process with a wall-clock limit: a vendor library that decides to tune or compile kernels on its first call
    (MIOpen's depthwise convolutions) must not be able to stretch the default bench run; the child builds its own CPU-oracle reference."""
    cmd = [sys.executable, os.path.abspath(__file__), "--vendor-baseline-only", "--arch", args.arch, "--img", str(args.img), "--batch", str(args.batch)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"error": f"child exited {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s} s (vendor library still tuning / compiling): skipped"}


def cpu_train_baseline(arch, img, batch=2, steps=1, warmup=0):
    """BASELINE configs[2] on the host: the oracle forward under torch autograd + the reference's criterion and optimizer
    (Charbonnier eps 1e-3, torch.optim.AdamW 2e-4 / 0.02), batch 2.  One step is ~15 s on 8 cores, so the default is ONE timed step
    without a warm-up step (the forward primitives are warm from cpu_baseline(), which runs first; thread count = its best one)."""
    from oracle import uformer_oracle as O
    from uformer_amd import spec
    cfg = spec.arch_config(arch, img_size=img)
    sd = spec.synth_state_dict(cfg, 1234)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    x, tgt = spec.synth_input(batch, img, img, 1234), spec.synth_input(batch, img, img, 1235)
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    times = []
    for i in range(steps + warmup):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = O.charbonnier_loss(O.uformer_forward(x, params, **kw), tgt)
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    v = batch / statistics.median(times)
    return {"value": v, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle forward under torch autograd + Charbonnier + torch.optim.AdamW, {arch} {img}x{img} batch {batch}, fp32: "
                       f"median of {steps} step(s) after {warmup} warm-up step(s) = {v:.3f} img/s ({sum(times):.0f} s of CPU work)")}


def train_mode(args, cfg, sd, dev, ud, dtype_name):
    """BASELINE configs[2]: forward + backward + Charbonnier + AdamW, batch `--train-batch` per GPU, train() mode (DropPath 0.1),
    every step on the native kernels (uformer_amd.train / losses / optim).  Returns the ``modes.train`` entry."""
    from uformer_amd import losses as ul
    from uformer_amd import model as um
    from uformer_amd import optim as uo
    from uformer_amd import spec
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    torch.manual_seed(1234 + rank)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()           # peak_mem_gb below is this leg's, not the inference modes' before it
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                   dd_in=cfg.dd_in, compute_dtype=TORCH_DTYPE[dtype_name])
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    opt = uo.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)     # train/train_denoise.py:77
    crit = ul.CharbonnierLoss()
    B = args.train_batch
    x = spec.synth_input(B, args.img, args.img, 1234 + 2 * rank).to(dev)
    tgt = spec.synth_input(B, args.img, args.img, 1235 + 2 * rank).to(dev)
    # f16 operands train under the reference's protocol (torch.cuda.amp.GradScaler through timm's NativeScaler, train/train_denoise.py:42, :180-184):
    # a DYNAMIC loss scale starting at 65536 with an inf / nan check and a skipped step on overflow -- uformer_amd.optim.GradScaler keeps all of it on
    # the device (no host synchronisation in the step); bf16 / f32 need no scale
    scaler = uo.GradScaler(device=dev) if dtype_name == "f16" else None
    ls = 65536.0 if dtype_name == "f16" else 1.0
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    sink = ud.OverlappedGradientAllReduce(m, algorithm=args.exchange, payload=TORCH_DTYPE[args.exchange_payload]) if world > 1 else None
    m.grad_sink = sink
    state = {}

    def step():
        if sink is not None:
            sink.begin_step()
        else:
            opt.zero_grad(set_to_none=True)
        loss = crit(m(x), tgt)
        (scaler.scale(loss) if scaler is not None else loss).backward()
        if sink is not None:
            # exposed exchange time: what the compute stream still waits for the collectives AFTER the last backward kernel (0 = fully hidden)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sink.finish()
            e1.record()
            state.setdefault("exchange_events", []).append((e0, e1))
        gsc = sink.grad_scale if sink is not None else 1.0
        if scaler is not None:
            scaler.step(opt, grad_scale=gsc)
            scaler.update()
        else:
            opt.step(grad_scale=gsc)
        state["loss"] = loss

    for _ in range(max(1, args.train_warmup)):
        step()
    torch.cuda.synchronize(); ud.barrier(); torch.cuda.synchronize()
    state.pop("exchange_events", None)
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        step()
    torch.cuda.synchronize(); ud.barrier()
    dt = ud.max_over_ranks(time.perf_counter() - t0, dev) / args.train_steps
    exposed_ms = None
    if sink is not None:
        ev = state.pop("exchange_events", [])
        exposed_ms = ud.max_over_ranks(sum(a_.elapsed_time(b_) for a_, b_ in ev) / max(1, len(ev)), dev)
    flops_img = 3 * 2.0 * m.flops()                        # FLOPs_train = 3 x forward
    v = world * B / dt
    out = {"workload": f"{args.arch} {args.img}x{args.img} training step (fwd + bwd + Charbonnier + AdamW), batch {B}/GPU, DropPath 0.1, synthetic data",
           "images_per_s": v, "ms_per_step": 1e3 * dt, "steps": args.train_steps, "batch_per_gpu": B, "dtype": dtype_name,
           "loss_scale": (scaler.get_scale() if scaler is not None else 1.0), "loss_scale_policy": ("dynamic (device-side GradScaler)" if scaler is not None else "none"),
           "loss_is_finite": bool(torch.isfinite(state["loss"]).item()),
           "loss": float(state["loss"].detach()), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
           "mfma_frac_whole_step": v * flops_img / 1e12 / world / MFMA_PEAK_TFLOPS[dtype_name],
           "gradient_exchange": "none (1 GPU)" if world == 1 else
           f"{'RCCL' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend()} "
           f"{'all-reduce' if args.exchange == 'ring' else 'direct exchange (all-to-all + f32 accumulation + all-gather, ' + args.exchange_payload + ' on the wire)'}, "
           f"{len(sink.buckets)} buckets overlapped with the reverse sweep",
           "global_batch": world * B, "exchange_buckets": (len(sink.buckets) if sink is not None else 0),
           "exchange_bytes_per_step": (sum(int(f_.numel()) * 4 for f_ in sink.flat) if sink is not None else 0),
           "exchange_wire_bytes_per_rank_per_step": (sink.wire_bytes_per_step() if sink is not None else 0),
           "exchange_exposed_ms_per_step": exposed_ms}
    # the instrumented step runs on EVERY rank (it contains the gradient exchange: rank 0 alone would wait for its peers forever); rank 0 reads its own report
    rows = kernel_breakdown(None, None, 1, fn=step)
    if rank == 0:
        sym = {}
        for r in rows:
            a_ = sym.setdefault(r["kernel"].split(" ")[0], {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
            for k_ in ("ms", "launches", "flops", "bytes"):
                a_[k_] += r[k_]
        bwd = {k: v_ for k, v_ in sym.items() if "wgrad" in k or "bwd" in k} or sym
        name, dom = max(bwd.items(), key=lambda kv: kv[1]["ms"])
        sec = dom["ms"] / 1e3
        ridge = MFMA_PEAK_TFLOPS[dtype_name] * 1e12 / (HBM_PEAK_GBS * 1e9)
        if dom["bytes"] > 0 and dom["flops"] / dom["bytes"] < ridge:
            ach = dom["bytes"] / sec / 1e9
            rf = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
        else:
            ach = dom["flops"] / sec / 1e12
            rf = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[dtype_name], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[dtype_name], "traffic": None}
        # measured HBM bytes per launch of that symbol (scripts/pmc_train.sh: separate FETCH_SIZE / WRITE_SIZE passes of the same step), if
        # they were taken on exactly these kernel sources
        for tf in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic_train.json")), reverse=True):
            try:
                tj = json.load(open(tf))
            except (OSError, ValueError):
                continue
            if tj.get("kernel_source_sha") == kernel_source_sha() and name in tj.get("kernels", {}) and dtype_name == "bf16" and B == 32:
                rf["traffic"] = tj["kernels"][name]["hbm_bytes_per_launch"]
                rf["traffic_note"] = (f"HBM bytes per launch, FETCH_SIZE x2 + WRITE_SIZE ({os.path.basename(tf)}); algorithmic bytes per launch "
                                      f"{dom['bytes'] / max(1, dom['launches']):.3e}; whole step: {tj.get('hbm_bytes_per_step_all_kernels', 0) / 1e9:.1f} GB")
                break
        rf.update({"kernel": name, "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / max(1, dom["launches"]),
                   "achieved_tflops": dom["flops"] / sec / 1e12, "achieved_gbs": dom["bytes"] / sec / 1e9,
                   "note": "dominant instrumented backward kernel of one training step (HIP events on the launch stream; the step runs single-stream while instrumented)",
                   "instrumented_gpu_ms_per_step": sum(v_["ms"] for v_ in sym.values())})
        out["roofline"] = rf
        out["kernels"] = [{"kernel": k_, "ms_per_step": v_["ms"], "launches_per_step": v_["launches"], "tflops": v_["flops"] / max(v_["ms"], 1e-9) / 1e9,
                           "gbs": v_["bytes"] / max(v_["ms"], 1e-9) / 1e6} for k_, v_ in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])[:12]]
    del m, opt
    torch.cuda.empty_cache()
    return out


def p720_mode(args, dev, ud, dtype_name, steps=3, warmup=1):
    """BASELINE configs[4]: ONE 1280x720 frame, padded to a 1280x1280 square as the reference's test script does (test/test_sidd.py:79-92,
    expand2square to a multiple of 128), through Uformer-B at that resolution, cropped back -- uformer_amd.infer.restore, every step on the
    native kernels.  Returns the ``modes.p720`` entry (synthetic frame resident in HBM; parity of this path: tests/test_gpu_model.py against
    the reference fixture model_B_720p.npz)."""
    from uformer_amd import infer, spec
    cfg = spec.arch_config(args.arch, img_size=256)
    sd = spec.synth_state_dict(cfg, 1234)
    m = build_model(args, cfg, sd, dev, TORCH_DTYPE[dtype_name])
    frame = spec.synth_input(1, 720, 1280, 4321).to(dev)
    with torch.no_grad():
        for _ in range(warmup):
            y = infer.restore(m, frame)
        torch.cuda.synchronize(); ud.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = infer.restore(m, frame)
        torch.cuda.synchronize(); ud.barrier()
        dt = ud.max_over_ranks(time.perf_counter() - t0, dev) / steps
    assert torch.isfinite(y).all() and tuple(y.shape[-2:]) == (720, 1280)
    flops = 2.0 * m.flops() * (1280 * 1280) / (256.0 * 256.0)      # the padded square is what runs
    out = {"workload": "Uformer_B, one 1280x720 frame -> expand2square 1280x1280 -> forward -> crop (uformer_amd.infer.restore), synthetic frame resident in HBM",
           "ms_per_frame": 1e3 * dt, "frames_per_s": 1.0 / dt, "steps": steps, "dtype": dtype_name,
           "mfma_frac": flops / dt / 1e12 / MFMA_PEAK_TFLOPS[dtype_name]}
    del m
    torch.cuda.empty_cache()
    return out


def p720_tiled_mode(args, dev, ud, dtype_name, steps=3, warmup=1, tile=768, min_overlap=128):
    """BASELINE configs[4] names a "window/overlap tiling path": the same 1280x720 frame through uformer_amd.infer.restore_tiled -- two 768x768 tiles
    with a 256-pixel linear-ramp overlap, forwarded as one batch (1.18 M pixels instead of the 1.64 M of the padded square) -- and its PSNR against
    the full-frame result of modes.p720 (the tiled path is an approximation by construction: a tile does not see what the whole frame sees; the
    reference itself always pads to the square, test/test_sidd.py:79-92,106-109)."""
    from oracle import uformer_oracle as O
    from uformer_amd import infer, spec
    cfg = spec.arch_config(args.arch, img_size=256)
    sd = spec.synth_state_dict(cfg, 1234)
    m = build_model(args, cfg, sd, dev, TORCH_DTYPE[dtype_name])
    frame = spec.synth_input(1, 720, 1280, 4321).to(dev)
    with torch.no_grad():
        full = infer.restore(m, frame)
        for _ in range(warmup):
            y = infer.restore_tiled(m, frame, tile=tile, min_overlap=min_overlap)
        torch.cuda.synchronize(); ud.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = infer.restore_tiled(m, frame, tile=tile, min_overlap=min_overlap)
        torch.cuda.synchronize(); ud.barrier()
        dt = ud.max_over_ranks(time.perf_counter() - t0, dev) / steps
    assert torch.isfinite(y).all() and tuple(y.shape[-2:]) == (720, 1280)
    ys, xs = infer._starts(720, tile, min_overlap), infer._starts(1280, tile, min_overlap)
    out = {"workload": f"Uformer_B, one 1280x720 frame -> {len(ys) * len(xs)} overlapped {tile}x{tile} tiles (min overlap {min_overlap}, linear ramps) in one batch -> blend "
                       "(uformer_amd.infer.restore_tiled), synthetic frame resident in HBM",
           "ms_per_frame": 1e3 * dt, "frames_per_s": 1.0 / dt, "steps": steps, "dtype": dtype_name, "tiles": len(ys) * len(xs),
           "pixels_forwarded_vs_padded_square": len(ys) * len(xs) * tile * tile / (1280.0 * 1280.0),
           "psnr_db_vs_full_frame": O.psnr(y.float().cpu(), full.float().cpu()), "max_abs_diff_vs_full_frame": float((y.float() - full.float()).abs().max())}
    del m
    torch.cuda.empty_cache()
    return out


def pipelined_mode(args, model, x, ud, dev, depth=2, small=((4, 4), (1, 4))):
    """The SAME forwards with up to ``depth`` of them in flight (uformer_amd.infer.PipelinedForward: successive batches on a ring of streams) -- a serving
    protocol, NOT the line's ``value`` (which stays one forward after the other on one stream, comparable with every earlier round): regions of exactly
    ``--steps`` forwards of ``--batch`` images between barrier + synchronize pairs, median of ``--repeats`` (at most 5), max over ranks; outputs compared
    bit for bit with the eager forward.  ``small``: (batch, depth) pairs of the small-batch regime the reference's evaluation scripts run in
    (test/test_sidd.py:101-107), each against its own eager loop, on rank 0's timing only."""
    from uformer_amd import infer, spec
    out = {"depth": depth, "protocol": f"up to {depth} forwards of successive batches in flight, each on its own stream; every handle consumed inside the timed region"}
    with torch.no_grad():
        want = model(x).clone()
        pf = infer.PipelinedForward(model, depth=depth)
        ys = list(pf.map([x] * (depth + 1)))
        torch.cuda.synchronize()
        out["bit_identical_to_eager"] = all(torch.equal(y, want) for y in ys)
        regions = []
        for _ in range(max(1, min(args.repeats, 5))):
            torch.cuda.synchronize(); ud.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _y in pf.map(x for _ in range(args.steps)):
                pass
            torch.cuda.synchronize(); ud.barrier()
            regions.append(ud.max_over_ranks(time.perf_counter() - t0, dev))
        e = statistics.median(regions)
        images = ud.sum_over_ranks(float(x.shape[0] * args.steps), dev)
        out.update({"images_per_s": images / e, "ms_per_step": 1e3 * e / args.steps, "steps": args.steps, "repeats": len(regions),
                    "images_per_s_min": images / max(regions), "images_per_s_max": images / min(regions)})
        sb = {}
        for (b, d) in small:
            xb = spec.synth_input(b, x.shape[2], x.shape[3], 4321).to(dev)
            n = max(args.steps, 160 // b)
            pfb = infer.PipelinedForward(model, depth=d)

            def loop_eager():
                for _ in range(n):
                    model(xb)

            def loop_piped():
                for _y in pfb.map(xb for _ in range(n)):
                    pass
            res = {}
            for name, fn in (("eager", loop_eager), ("pipelined", loop_piped)):
                fn()
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                res[name] = b * n / statistics.median(ts)
            sb[f"batch{b}"] = {"depth": d, "eager_images_per_s": res["eager"], "pipelined_images_per_s": res["pipelined"], "forwards_per_region": n}
        out["small_batch"] = sb
    return out


def build_model(args, cfg, sd, dev, cd):
    from uformer_amd import model as um
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                   modulator=cfg.modulator, dd_in=cfg.dd_in, compute_dtype=cd).eval()
    m.load_state_dict(sd, strict=True)
    return m.to(dev)


def timed_steps(model, x, steps, warmup, ud, dev, repeats=1):
    """`warmup` untimed forwards, then `repeats` timed regions of EXACTLY `steps` forwards each, every region bracketed by barrier + synchronize on both
    sides and reduced with MAX over ranks.  Returns the list of region times (seconds).  One region of 20 steps is 0.13 s of GPU time: the spread between
    regions (and between boxes, +-3-4 %) is larger than most single kernel changes, so the line reports the MEDIAN region with min / max."""
    out = []
    with torch.no_grad():
        for _ in range(warmup):
            y = model(x)
        for _ in range(max(1, repeats)):
            torch.cuda.synchronize()
            ud.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = model(x)
            torch.cuda.synchronize()
            ud.barrier()
            t1 = time.perf_counter()
            out.append(ud.max_over_ranks(t1 - t0, dev))
    assert torch.isfinite(y).all()
    return out


def launcher_argv(n_gpus: int, port: int, bench_args):
    """The command ``python bench.py --gpus N`` turns itself into when no torchrun environment is present: one process per GPU
    under torch.distributed.run on this node, rendezvous on the loopback address (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(bench_args)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def needs_self_launch(n_gpus: int, env) -> bool:
    """True for ``python bench.py --gpus N`` (N > 1) started WITHOUT torch.distributed.run: no RANK / WORLD_SIZE in the environment."""
    return n_gpus > 1 and "RANK" not in env and int(env.get("WORLD_SIZE", "1")) <= 1


def self_launch(n_gpus: int, bench_args, device_count=None) -> int:
    """Re-run this script as N ranks (one per GPU); returns the launcher's exit code.  A box with fewer than N GPUs is a clear
    error, not a hang at the rendezvous."""
    have = torch.cuda.device_count() if device_count is None else device_count
    if os.environ.get("UF_BENCH_SHARE_GPU") == "1" and device_count is None:
        have = max(have, n_gpus) if have >= 1 else have       # functional test of the N-rank path on a 1-GPU box (see main): ranks share device 0
    if have < n_gpus:
        raise SystemExit(f"bench.py --gpus {n_gpus}: this node exposes {have} GPU(s) to this process "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}); run with --gpus <= {max(have, 1)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    return subprocess.call(launcher_argv(n_gpus, free_port(), bench_args), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="timed regions of --steps forwards each; value = the median region (min / max reported beside it)")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--arch", default="Uformer_B")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vendor-baseline", action="store_true", help="skip the same-node PyTorch-ROCm eager calibration (oracle/vendor_forward.py)")
    ap.add_argument("--no-f32-mode", action="store_true", help="skip the exact-f32 mode")
    ap.add_argument("--no-other-modes", action="store_true", help="headline mode only (no f16 / bf16 / f32 companions)")
    ap.add_argument("--no-train-mode", action="store_true", help="skip modes.train (BASELINE configs[2])")
    ap.add_argument("--no-720p", action="store_true", help="skip modes.p720 (BASELINE configs[4]: one 1280x720 frame through expand2square -> 1280x1280)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip modes.pipelined (two forwards of successive batches in flight; small-batch throughput)")
    ap.add_argument("--train-mode-multi", action="store_true", help="(default since round 5; kept for old command lines) run modes.train under N > 1 too")
    ap.add_argument("--train-batch", type=int, default=32)
    ap.add_argument("--train-steps", type=int, default=3)
    ap.add_argument("--train-warmup", type=int, default=1)
    ap.add_argument("--train-dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--exchange", default="ring", choices=["ring", "direct"], help="gradient exchange of modes.train under N > 1: one all-reduce per bucket (ring: the library's "
                    "schedule) or the direct two-step form over all xGMI links (uformer_amd.dist.OverlappedGradientAllReduce)")
    ap.add_argument("--exchange-payload", default="f32", choices=["f32", "bf16"], help="wire type of --exchange direct (accumulation is f32 either way)")
    ap.add_argument("--error-budget", action="store_true", help="bf16-mode error by source through oracle/bf16_budget.py (about a CPU-minute)")
    ap.add_argument("--kernels-json", default=None, help="also write the per-kernel breakdown to this file")
    ap.add_argument("--vendor-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.vendor_baseline_only:                                # child of vendor_baseline_guarded(): one JSON line, nothing else
        from oracle import uformer_oracle as O
        from uformer_amd import spec
        cfg_ = spec.arch_config(args.arch, img_size=args.img)
        x1_ = spec.synth_input(1, args.img, args.img, 1234)
        with torch.no_grad():
            ref_ = O.uformer_forward(x1_, spec.synth_state_dict(cfg_, 1234), img_size=cfg_.img_size, embed_dim=cfg_.embed_dim, depths=cfg_.depths,
                                     num_heads=cfg_.num_heads, dd_in=cfg_.dd_in)
        print(json.dumps(vendor_baseline(args.arch, args.img, args.batch, torch.device("cuda", 0), x1_, ref_)), flush=True)
        return
    if needs_self_launch(args.gpus, os.environ):               # `python bench.py --gpus N`: become N ranks under torch.distributed.run
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    from uformer_amd import dist as ud
    from uformer_amd import spec

    # UF_BENCH_BACKEND=gloo UF_BENCH_SHARE_GPU=1: functional test of the N-rank code path (launcher, sharding, gradient sink, JSON assembly) on a box with ONE
    # GPU -- every rank uses device 0 and the collectives go through gloo; the numbers of such a run mean nothing and the line says so
    backend = os.environ.get("UF_BENCH_BACKEND", "nccl")
    share = os.environ.get("UF_BENCH_SHARE_GPU") == "1"
    rank, local_rank, world = ud.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cd = TORCH_DTYPE[args.dtype]

    cfg = spec.arch_config(args.arch, img_size=args.img)
    sd = spec.synth_state_dict(cfg, 1234)
    model = build_model(args, cfg, sd, dev, cd)
    # weak scaling: every rank owns `batch` images of the global batch world*batch
    a, b = ud.shard_batch(args.batch * world, rank, world)
    x = spec.synth_input(b - a, args.img, args.img, 1234 + rank).to(dev)

    devices = ud.rank_devices(dev)          # gathered over the process group (RCCL): one row per rank that actually met
    regions = timed_steps(model, x, args.steps, args.warmup, ud, dev, repeats=args.repeats)
    elapsed = statistics.median(regions)
    images = ud.sum_over_ranks(float((b - a) * args.steps), dev)
    # the other operand types: same steps for the 2-byte types, fewer for exact-f32 MFMA (1/16 of the bf16 / f16 rate)
    others = {}
    if not args.no_other_modes and args.arch == "Uformer_B":
        for other in [d for d in ("bf16", "f16", "f32") if d != args.dtype and not (d == "f32" and args.no_f32_mode)]:
            steps2 = max(3, args.steps // 4) if other == "f32" else args.steps
            m2 = build_model(args, cfg, sd, dev, TORCH_DTYPE[other])
            r2 = timed_steps(m2, x, steps2, 2, ud, dev, repeats=1 if other == "f32" else min(args.repeats, 5))
            others[other] = (m2, statistics.median(r2), steps2, ud.sum_over_ranks(float((b - a) * steps2), dev), r2)
    pipe_entry = None
    if not args.no_pipelined and not args.no_other_modes:
        try:
            pipe_entry = pipelined_mode(args, model, x, ud, dev)
        except Exception as e:      # noqa: BLE001 -- a companion mode must not take the headline line down (single process); under N > 1 its peers sit in collectives: stop
            import traceback
            print(f"[bench rank {rank}] modes.pipelined failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            pipe_entry = {"error": f"{type(e).__name__}: {e}"[:500]}
            if world > 1:
                raise
    train_entry = None
    # modes.train rides along at every N: under N > 1 it is the one place of the path with a collective (the bucketed gradient all-reduce
    # over RCCL, overlapped with the reverse sweep), so a scaling run exercises it by default (--no-train-mode skips it)
    if not args.no_train_mode and args.arch == "Uformer_B":
        try:
            train_entry = train_mode(args, cfg, sd, dev, ud, args.train_dtype)
        except Exception as e:      # noqa: BLE001 -- the headline (inference) line must survive a failure of the companion mode; the error is reported in it
            import traceback
            print(f"[bench rank {rank}] modes.train failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            train_entry = {"error": f"{type(e).__name__}: {e}"[:500]}
            torch.cuda.empty_cache()
            if world > 1:
                # a rank that failed alone has left its peers inside train_mode's collectives (gradient buckets, barrier, max over ranks): the collective
                # sequences of the ranks no longer match and the inference legs below would hang in RCCL instead of surviving -- stop here
                raise

    p720_entry = p720t_entry = None
    if not args.no_720p and args.arch == "Uformer_B" and args.img == 256:
        p720_entry = p720_mode(args, dev, ud, args.dtype)
        p720t_entry = p720_tiled_mode(args, dev, ud, args.dtype)

    out = None
    if rank == 0:
        value = images / elapsed
        flops_img = 2.0 * model.flops()   # exact MACs of this arch at the constructor resolution
        out = {
            "metric": "restored images/sec (256x256, Uformer-B)", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # value = the MEDIAN of `repeats` timed regions of exactly `steps` forwards each (every region between barrier + synchronize pairs, max over ranks)
            "repeats": len(regions), "value_min": images / max(regions), "value_max": images / min(regions),
            "region_ms_per_step": [round(1e3 * r_ / args.steps, 4) for r_ in regions],
            "config": {"workload": f"{args.arch} {args.img}x{args.img} inference, batch {args.batch}/GPU, "
                                   f"synthetic U[0,1) images resident in HBM, synthetic trained-like weights"
                                   + ("; bf16 is the operand type BASELINE.json configs[1] names -- it misses the 1e-3 north-star tolerance BY DESIGN (rounding the weights alone to bf16 "
                                      "costs 1.3e-3, oracle/bf16_budget.py); value_at_tolerance = the fastest mode that meets 1e-3 (f16, the reference's own AMP type)" if args.dtype == "bf16" else ""),
                       "global_batch": args.batch * world, "parallelism": f"batch-sharded replicas x{world}, no collective",
                       "ranks": world, "world_size_reported_by_process_group": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       "rank_devices": [{"rank": r[0], "local_rank": r[1], "device_index": r[2], "device_id_hash": r[3]} for r in devices],
                       "collective_backend": ("RCCL (torch.distributed nccl backend): barrier + max/sum of the timings + the rank_devices all-gather only" if backend == "nccl" else
                                              f"{backend} (FUNCTIONAL TEST of the N-rank path{', ranks share one GPU' if share else ''}: not a scaling measurement)")},
            "model_gflop_per_image": flops_img / 1e9,
            "mfma_frac_whole_model": value * flops_img / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype],
            "kernel_source_sha": kernel_source_sha(),
        }
        if args.arch == "Uformer_B" and args.img == 256 and args.dtype == "bf16":
            # SURVEY 8d "report both fractions": compulsory (ideal whole-block fusion) HBM bytes vs the 8 TB/s peak
            out["hbm_frac_whole_model_compulsory"] = value * (MB_PER_IMAGE_B256 + MB_WEIGHTS_B / args.batch) * 1e6 / world / (HBM_PEAK_GBS * 1e9)
        out["modes"] = {args.dtype: {"images_per_s": value, "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps,
                                     "mfma_frac_whole_model": out["mfma_frac_whole_model"]}}
        for other, (m2, e2, steps2, images2, r2) in others.items():
            v2 = images2 / e2
            out["modes"][other] = {"images_per_s": v2, "ms_per_step": 1e3 * e2 / steps2, "steps": steps2, "repeats": len(r2),
                                   "images_per_s_min": images2 / max(r2), "images_per_s_max": images2 / min(r2),
                                   "mfma_frac_whole_model": v2 * flops_img / 1e12 / world / MFMA_PEAK_TFLOPS[other]}
        if train_entry is not None:
            out["modes"]["train"] = train_entry
        if pipe_entry is not None:
            out["modes"]["pipelined"] = pipe_entry
        if p720_entry is not None:
            out["modes"]["p720"] = p720_entry
        if p720t_entry is not None:
            out["modes"]["p720_tiled"] = p720t_entry
        # ---- roofline of the dominant kernel: HIP events on the launch stream, per kernel class ----
        rows = kernel_breakdown(model, x, 3)
        total_ms = sum(r["ms"] for r in rows)
        # aggregate per kernel SYMBOL (the name before the shape suffix) -- the granularity rocprofv3 reports
        sym = {}
        for r in rows:
            a_ = sym.setdefault(r["kernel"].split(" ")[0], {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
            for k_ in ("ms", "launches", "flops", "bytes"):
                a_[k_] += r[k_]
        name, dom = max(sym.items(), key=lambda kv: kv[1]["ms"])
        sec = dom["ms"] / 1e3
        # bound by arithmetic intensity of the kernel's ALGORITHMIC work vs the machine ridge (flop/byte)
        ridge = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
        if dom["bytes"] <= 0 or dom["flops"] / dom["bytes"] >= ridge:
            ach = dom["flops"] / sec / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                               "frac": ach / MFMA_PEAK_TFLOPS[args.dtype], "traffic": None}
        else:
            ach = dom["bytes"] / sec / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None}
        # HBM bytes per launch of that kernel from the PMC passes of scripts/official_run.sh -- only if they were collected on
        # exactly these kernel sources (stamp); a number from an older build would silently go stale, so it is dropped instead
        tfiles = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True)       # newest round first
        note = "null: no PMC passes of this build (scripts/official_run_r06.sh collects them)"
        for tf in tfiles:
            try:
                with open(tf) as f:
                    tj = json.load(f)
            except (OSError, ValueError):
                continue
            if tj.get("kernel_source_sha") != out["kernel_source_sha"]:
                note = f"null: {os.path.basename(tf)} was measured on other kernel sources (stamp mismatch)"
                continue
            tr = tj.get("kernels", {}).get(name)
            if tr:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                # NOT measured in this run: rocprofv3 --pmc passes cannot run inside the timed process; the number is REPLAYED from the committed file below,
                # which scripts/official_run_r06.sh wrote on a build with exactly this kernel_source_sha (a mismatch drops it to null)
                out["roofline"]["traffic_source"] = f"replayed from profiles/{os.path.basename(tf)} (same kernel_source_sha; separate rocprofv3 --pmc passes)"
                note = ("HBM bytes per launch, FETCH_SIZE x2 + WRITE_SIZE (separate --pmc passes of this workload, same "
                        f"kernel sources: profiles/{os.path.basename(tf)}); algorithmic bytes per launch {dom['bytes'] / dom['launches']:.4g}")
                if "mfma_busy_frac" in tr:         # the SQ passes of the same run: share of the SIMD-cycles with the MFMA pipe / the VALU busy (scripts/pmc_traffic.py)
                    out["roofline"]["mfma_busy_frac"] = tr["mfma_busy_frac"]
                    out["roofline"]["valu_active_frac"] = tr["valu_active_frac"]
            break
        out["roofline"]["traffic_note"] = note
        out["roofline"]["flop_per_byte"] = dom["flops"] / max(dom["bytes"], 1.0)
        out["roofline"]["achieved_tflops"] = dom["flops"] / sec / 1e12
        out["roofline"].update({"kernel": name, "launches_per_step": dom["launches"] // 3,
                                "avg_launch_ms": dom["ms"] / dom["launches"], "share_of_gpu_time": dom["ms"] / total_ms,
                                "gpu_ms_per_step_all_kernels": total_ms / 3})
        # the same roofline arithmetic for the three largest symbols (the dominant one changes hands between attn_block<512> and <256,256>
        # from box to box: they are within 3 % of each other), so a reader can follow ONE kernel across rounds
        def _roof(v_):
            sec_ = v_["ms"] / 1e3
            if v_["bytes"] <= 0 or v_["flops"] / v_["bytes"] >= ridge:
                a_ = v_["flops"] / sec_ / 1e12
                return {"bound": "mfma", "achieved": a_, "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": a_ / MFMA_PEAK_TFLOPS[args.dtype]}
            a_ = v_["bytes"] / sec_ / 1e9
            return {"bound": "hbm", "achieved": a_, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_ / HBM_PEAK_GBS}
        out["roofline_top3"] = [dict(_roof(v_), kernel=k_, avg_launch_ms=v_["ms"] / v_["launches"], share_of_gpu_time=v_["ms"] / total_ms,
                                     mfma_frac=v_["flops"] / (v_["ms"] / 1e3) / 1e12 / MFMA_PEAK_TFLOPS[args.dtype])
                                for k_, v_ in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])[:3]]
        out["kernels"] = [{"kernel": k_, "ms_per_step": v_["ms"] / 3, "launches_per_step": v_["launches"] // 3,
                           "tflops": v_["flops"] / max(v_["ms"], 1e-9) / 1e9, "gbs": v_["bytes"] / max(v_["ms"], 1e-9) / 1e6}
                          for k_, v_ in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])]
        if args.kernels_json:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernels_json)), exist_ok=True)
            with open(args.kernels_json, "w") as f:
                json.dump(rows, f, indent=1)
        # ---- CPU baseline (oracle, host cores) + parity of image 0 against it, every mode -----------------------
        if world == 1 and not args.no_cpu_baseline:
            cb, (x1, ref) = cpu_baseline(args.arch, args.img)
            out["cpu_baseline"] = cb
            from oracle import uformer_oracle as O
            out["parity"] = {"checked": "1 image, same weights/input as the CPU baseline, vs oracle/uformer_oracle.py (pinned to the reference's fixtures)",
                             "north_star_tolerance_max_abs": 1e-3}
            for mname, m in [(args.dtype, model)] + [(o, v_[0]) for o, v_ in others.items()]:   # noqa: E501
                with torch.no_grad():
                    y1 = m(x1.to(dev)).float().cpu()
                d = y1 - ref
                out["parity"][mname] = {"max_abs_err_vs_oracle": float(d.abs().max()), "mean_abs_err": float(d.abs().mean()), "mean_signed_err": float(d.mean()),
                                        "psnr_db_vs_oracle": O.psnr(y1, ref), "meets_1e-3": bool(d.abs().max() <= 1e-3)}
                out["modes"][mname].update(out["parity"][mname])
            ok_modes = [(out["modes"][mn]["images_per_s"], mn) for mn in ("bf16", "f16", "f32") if out["modes"].get(mn, {}).get("meets_1e-3")]
            if ok_modes:       # the number that satisfies BOTH halves of the north star: throughput AND <= 1e-3 max-abs against the reference restatement
                out["value_at_tolerance"] = max(ok_modes)[0]
                out["value_at_tolerance_mode"] = max(ok_modes)[1]
            out["parity"]["max_abs_err_vs_oracle"] = out["parity"][args.dtype]["max_abs_err_vs_oracle"]
            out["parity"]["psnr_db_vs_oracle"] = out["parity"][args.dtype]["psnr_db_vs_oracle"]
            if not args.no_vendor_baseline:
                out["vendor_baseline"] = vendor_baseline_guarded(args)
            if train_entry is not None and "error" not in train_entry:
                train_entry["cpu_baseline"] = cpu_train_baseline(args.arch, args.img)
            if args.error_budget:
                from oracle import bf16_budget as BB
                from uformer_amd import spec as sp
                c2 = sp.arch_config(args.arch, img_size=args.img)
                for op_ in ("bf16", "f16"):
                    out["parity"]["error_budget_" + op_] = {
                        "method": f"oracle/bf16_budget.py (operand={op_}): the f32 oracle with ONE rounding point / approximation of the {op_} kernels switched on at a time",
                        "by_source": BB.error_budget(x1, sd, ref, img_size=c2.img_size, embed_dim=c2.embed_dim, depths=c2.depths, num_heads=c2.num_heads, dd_in=c2.dd_in,
                                                     operand=op_)}
        # ---- compact scalar block, LAST in the line (a log tail keeps it): every headline number of the modes above
        md = out["modes"]
        sm = {"dtype": args.dtype, "img_s": round(value, 1), "img_s_min": round(out["value_min"], 1), "img_s_max": round(out["value_max"], 1), "repeats": out["repeats"],
              "img_s_at_tolerance": (round(out["value_at_tolerance"], 1) if "value_at_tolerance" in out else None), "mode_at_tolerance": out.get("value_at_tolerance_mode"), "ms_step": round(1e3 * elapsed / args.steps, 3), "mfma_frac_model": round(out["mfma_frac_whole_model"], 4),
              "dom_kernel": out["roofline"]["kernel"], "dom_bound": out["roofline"]["bound"], "dom_frac": round(out["roofline"]["frac"], 4),
              "dom_avg_us": round(1e3 * out["roofline"]["avg_launch_ms"], 1), "dom_traffic": out["roofline"]["traffic"],
              "dom_mfma_busy": (round(out["roofline"]["mfma_busy_frac"], 3) if "mfma_busy_frac" in out["roofline"] else None),
              "dom_valu_active": (round(out["roofline"]["valu_active_frac"], 3) if "valu_active_frac" in out["roofline"] else None),
              "gpu_ms_step_kernels": round(out["roofline"]["gpu_ms_per_step_all_kernels"], 3)}
        for mname in ("bf16", "f16", "f32"):
            if mname in md:
                sm[mname + "_img_s"] = round(md[mname]["images_per_s"], 1)
                if "max_abs_err_vs_oracle" in md[mname]:
                    sm[mname + "_err"] = float("%.3e" % md[mname]["max_abs_err_vs_oracle"])
                    sm[mname + "_meets_1e-3"] = md[mname]["meets_1e-3"]
        if "train" in md and "error" in md["train"]:
            sm["train_error"] = md["train"]["error"][:120]
        elif "train" in md:
            tr = md["train"]
            sm.update({"train_img_s": round(tr["images_per_s"], 1), "train_ms_step": round(tr["ms_per_step"], 2), "train_dtype": tr["dtype"], "train_batch": tr["batch_per_gpu"],
                       "train_mfma_frac": round(tr["mfma_frac_whole_step"], 4), "train_peak_mem_gb": round(tr["peak_mem_gb"], 1),
                       "train_global_batch": tr["global_batch"], "train_exchange_buckets": tr["exchange_buckets"],
                       "train_exchange_exposed_ms": (round(tr["exchange_exposed_ms_per_step"], 3) if tr["exchange_exposed_ms_per_step"] is not None else None)})
            if "roofline" in tr:
                sm.update({"train_dom_kernel": tr["roofline"]["kernel"], "train_dom_bound": tr["roofline"]["bound"], "train_dom_frac": round(tr["roofline"]["frac"], 4),
                           "train_dom_traffic": tr["roofline"].get("traffic")})
            if "cpu_baseline" in tr:
                sm["train_cpu_img_s"] = round(tr["cpu_baseline"]["value"], 3)
        if "pipelined" in md and "error" in md["pipelined"]:
            sm["pipelined_error"] = md["pipelined"]["error"][:120]
        elif "pipelined" in md:
            pp = md["pipelined"]
            sm.update({"pipelined_depth": pp["depth"], "pipelined_img_s": round(pp["images_per_s"], 1), "pipelined_bit_identical": pp["bit_identical_to_eager"]})
            for k_, v_ in pp.get("small_batch", {}).items():
                sm[k_ + "_img_s"] = round(v_["eager_images_per_s"], 1)
                sm[k_ + f"_pipelined{v_['depth']}_img_s"] = round(v_["pipelined_images_per_s"], 1)
        if "p720" in md:
            sm.update({"p720_ms": round(md["p720"]["ms_per_frame"], 2), "p720_fps": round(md["p720"]["frames_per_s"], 1), "p720_mfma_frac": round(md["p720"]["mfma_frac"], 4)})
        if "p720_tiled" in md:
            sm.update({"p720_tiled_ms": round(md["p720_tiled"]["ms_per_frame"], 2), "p720_tiled_fps": round(md["p720_tiled"]["frames_per_s"], 1),
                       "p720_tiled_psnr_vs_full_db": round(md["p720_tiled"]["psnr_db_vs_full_frame"], 1)})
        if "vendor_baseline" in out:
            for k_ in ("fp32", "bf16_autocast", "f16_autocast"):
                if "images_per_s" in out["vendor_baseline"].get(k_, {}):
                    sm["vendor_" + k_ + "_img_s"] = round(out["vendor_baseline"][k_]["images_per_s"], 1)
        if "cpu_baseline" in out:
            sm.update({"cpu_img_s": round(out["cpu_baseline"]["value"], 3), "cpu_cores": out["cpu_baseline"]["cores"], "cpu_kind": out["cpu_baseline"]["kind"]})
        out["summary"] = sm
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
