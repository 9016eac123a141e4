#!/usr/bin/env python3
"""Headline benchmark: restored images/s, Uformer-B 256x256, bf16 inference, batch 16 per GPU
(BASELINE.json configs[1]).  One process per GPU; the batch is sharded across ranks with no
data-path collective (weak scaling: per-GPU work is fixed).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of the hot path (Uformer.forward through uf_uformer_fwd) over one batch of
synthetic images already resident in HBM.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks (same guide)
GFLOP_PER_IMAGE_B256 = 173.15   # 2 x 86.574 GMAC, Uformer-B @256x256
MB_PER_IMAGE_B256 = 184.8       # same section: compulsory bf16 activation bytes per image (blocks in+out once, samplers, skips, stem/head)
MB_WEIGHTS_B = 101.8            # bf16 weights, read once per batch


def kernel_breakdown(model, x, steps):
    """Per-kernel-class time over `steps` forwards, measured with HIP events on the launch stream
    (library-side instrumentation, uf_timing_enable)."""
    from uformer_amd import _lib
    lib = _lib.load()
    torch.cuda.synchronize()
    lib.uf_timing_enable(1)
    with torch.no_grad():
        for _ in range(steps):
            model(x)
    torch.cuda.synchronize()
    lib.uf_timing_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.uf_timing_report(buf, len(buf))
    rows = json.loads(buf.value.decode())
    for r in rows:
        r["ms_per_launch"] = r["ms"] / max(1, r["launches"])
    rows.sort(key=lambda r: -r["ms"])
    return rows


def cpu_baseline(arch, img, seconds=12.0):
    """The oracle (CPU restatement of the reference forward, fp32, eval) timed on this host."""
    from oracle import uformer_oracle as O
    from uformer_amd import spec
    cfg = spec.arch_config(arch, img_size=img)
    sd = spec.synth_state_dict(cfg, 1234)
    x = spec.synth_input(1, img, img, 1234)
    # torch's own default thread count: respects the container's CPU affinity / quota, unlike os.cpu_count()
    kw = dict(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
    with torch.no_grad():
        ref = O.uformer_forward(x, sd, **kw)      # warm-up, also the parity reference
        n, t0 = 0, time.perf_counter()
        while True:
            O.uformer_forward(x, sd, **kw)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 12:
                break
    return {"value": n / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} fp32 forwards of 1 image ({arch} {img}x{img}) through oracle/uformer_oracle.py in {dt:.1f} s"}, (x, ref)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--arch", default="Uformer_B")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels-json", default=None, help="also write the per-kernel breakdown to this file")
    args = ap.parse_args()

    from uformer_amd import dist as ud
    from uformer_amd import model as um
    from uformer_amd import spec

    rank, local_rank, world = ud.init_process_group("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    cfg = spec.arch_config(args.arch, img_size=args.img)
    sd = spec.synth_state_dict(cfg, 1234)
    model = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                       modulator=cfg.modulator, dd_in=cfg.dd_in, compute_dtype=cd).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    # weak scaling: every rank owns `batch` images of the global batch world*batch
    a, b = ud.shard_batch(args.batch * world, rank, world)
    x = spec.synth_input(b - a, args.img, args.img, 1234 + rank).to(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            y = model(x)
        torch.cuda.synchronize()
        ud.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x)
        torch.cuda.synchronize()
        ud.barrier()
        t1 = time.perf_counter()
    elapsed = ud.max_over_ranks(t1 - t0, dev)
    images = ud.sum_over_ranks(float((b - a) * args.steps), dev)
    assert torch.isfinite(y).all()

    out = None
    if rank == 0:
        value = images / elapsed
        flops_img = 2.0 * model.flops()   # exact MACs of this arch at the constructor resolution
        out = {
            "metric": "restored images/sec (256x256, Uformer-B)", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.arch} {args.img}x{args.img} inference, batch {args.batch}/GPU, "
                                   f"synthetic U[0,1) images resident in HBM, synthetic trained-like weights",
                       "global_batch": args.batch * world, "parallelism": f"batch-sharded replicas x{world}, no collective"},
            "model_gflop_per_image": flops_img / 1e9,
            "mfma_frac_whole_model": value * flops_img / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype],
        }
        if args.arch == "Uformer_B" and args.img == 256 and args.dtype == "bf16":
            # SURVEY 8d "report both fractions": compulsory (ideal whole-block fusion) HBM bytes vs the 8 TB/s peak
            out["hbm_frac_whole_model_compulsory"] = value * (MB_PER_IMAGE_B256 + MB_WEIGHTS_B / args.batch) * 1e6 / world / (HBM_PEAK_GBS * 1e9)
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
        # HBM bytes per launch of that kernel from the committed PMC passes of this same workload (None if never profiled)
        try:
            with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
                tr = json.load(f)["kernels"].get(name)
            if tr:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_note"] = ("bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/r01_pmc_traffic.json; "
                                                   f"algorithmic bytes per launch {dom['bytes'] / dom['launches']:.4g}")
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"]["flop_per_byte"] = dom["flops"] / max(dom["bytes"], 1.0)
        out["roofline"]["achieved_tflops"] = dom["flops"] / sec / 1e12
        out["roofline"].update({"kernel": name, "launches_per_step": dom["launches"] // 3,
                                "avg_launch_ms": dom["ms"] / dom["launches"], "share_of_gpu_time": dom["ms"] / total_ms,
                                "gpu_ms_per_step_all_kernels": total_ms / 3})
        out["kernels"] = [{"kernel": k_, "ms_per_step": v_["ms"] / 3, "launches_per_step": v_["launches"] // 3,
                           "tflops": v_["flops"] / max(v_["ms"], 1e-9) / 1e9, "gbs": v_["bytes"] / max(v_["ms"], 1e-9) / 1e6}
                          for k_, v_ in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])]
        if args.kernels_json:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernels_json)), exist_ok=True)
            with open(args.kernels_json, "w") as f:
                json.dump(rows, f, indent=1)
        # ---- CPU baseline (oracle, host cores) + parity of image 0 against it -----------------------
        if world == 1 and not args.no_cpu_baseline:
            cb, (x1, ref) = cpu_baseline(args.arch, args.img)
            out["cpu_baseline"] = cb
            with torch.no_grad():
                y1 = model(x1.to(dev)).float().cpu()
            from oracle import uformer_oracle as O
            out["parity"] = {"max_abs_err_vs_oracle": float((y1 - ref).abs().max()), "psnr_db_vs_oracle": O.psnr(y1, ref),
                             "checked": "1 image, same weights/input as the CPU baseline"}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
