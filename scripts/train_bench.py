#!/usr/bin/env python3
"""First measurement of BASELINE configs[2] (Uformer-B 256x256 training step: forward + backward + AdamW) on the op-by-op
backward of uformer_amd/train.py.  Not the contract bench (bench.py measures the inference metric); prints one JSON line.

    python scripts/train_bench.py --batch 32 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_bench.py --batch 32
        (BASELINE configs[3]: 32 images per GPU, gradients averaged by uformer_amd.dist.GradientAllReduce over RCCL;
         not yet run on a multi-GPU box)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uformer_amd import dist as ud  # noqa: E402
from uformer_amd import losses as ul  # noqa: E402
from uformer_amd import model as um  # noqa: E402
from uformer_amd import optim as uo  # noqa: E402
from uformer_amd import spec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--arch", default="Uformer_B")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--loss-scale", type=float, default=0.0, help="static loss scale (default: 65536 for f16 = GradScaler's initial scale, 1 otherwise)")
    ap.add_argument("--kernels-json", default=None, help="also run ONE instrumented step (HIP events around every library launch) and write the per-(kernel, shape) table here")
    ap.add_argument("--graph", action="store_true", help="capture forward + backward of one step in a HIP graph (torch.cuda.CUDAGraph) and replay it; AdamW stays outside (its step count is a launch argument)")
    ap.add_argument("--no-fused-attn", action="store_true", help="A/B: the op-by-op attention half of the kept-intermediates forward (round-5 form) instead of uf_lewin_attn_train_fwd")
    ap.add_argument("--sink", action="store_true", help="use the bucket-view gradient sink on one GPU too (exercises the overlapped path without a collective)")
    a = ap.parse_args()
    if a.no_fused_attn:
        from uformer_amd import train as _train
        _train._FUSED_ATTN_FWD = False
    rank, local_rank, world = ud.init_process_group("nccl")
    torch.cuda.set_device(local_rank)
    torch.manual_seed(1234 + rank)                                                 # DropPath masks differ per rank (train/train_denoise.py:60-63 seeds 1234)
    cfg = spec.arch_config(a.arch, img_size=a.img)
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                   dd_in=cfg.dd_in, compute_dtype={"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype])
    ls = a.loss_scale if a.loss_scale > 0 else (65536.0 if a.dtype == "f16" else 1.0)
    m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
    m = m.cuda().train()
    opt = uo.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)     # train/train_denoise.py:77, on uf_adamw_step
    criterion = ul.CharbonnierLoss()                                                          # losses.py:41-52, on uf_charbonnier_fwd_bwd
    x = spec.synth_input(a.batch, a.img, a.img, 1234 + 2 * rank).cuda()            # per-GPU batch (weak scaling)
    target = spec.synth_input(a.batch, a.img, a.img, 1235 + 2 * rank).cuda()
    # gradients are views into flat all-reduce buckets; the reverse sweep hands every finished stage to the sink, which launches
    # that bucket's RCCL all-reduce behind the producing kernels; 1/world is folded into the AdamW update
    sink = ud.OverlappedGradientAllReduce(m) if (world > 1 or a.sink) else None
    m.grad_sink = sink

    def step():
        if sink is not None:
            sink.begin_step()
        else:
            opt.zero_grad(set_to_none=True)
        loss = criterion(m(x), target)
        (loss * ls if ls != 1.0 else loss).backward()          # f16 operands: scaled loss; 1 / scale is folded into the AdamW update
        if sink is not None:
            sink.finish()
        opt.step(grad_scale=(sink.grad_scale if sink is not None else 1.0) / ls)
        return loss

    devices = ud.rank_devices("cuda")                                              # all-gathered over RCCL: proof that `world` ranks met
    if a.graph:
        # static inputs, a few eager steps on a side stream (allocator and library warm-up, every kernel's one-time attribute calls), then
        # forward + backward captured once: the gradients live in the graph's private pool and every replay rewrites them in place
        assert sink is None, "--graph: single-GPU form (the collective of the sink is not captured)"
        ws = torch.cuda.Stream()
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(ws)
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = criterion(m(x), target)
            (static_loss * ls if ls != 1.0 else static_loss).backward()

        def step():                                                                # noqa: F811
            graph.replay()
            opt.step(grad_scale=1.0 / ls)
            return static_loss
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    ud.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / a.steps * 1e3                            # the Python loop alone (enqueue); close to ms_per_step = host-bound
    torch.cuda.synchronize()
    ud.barrier()
    dt = ud.max_over_ranks(time.perf_counter() - t0, "cuda") / a.steps
    if rank == 0:
        print(json.dumps({"metric": "training images/sec (fused fwd + recompute bwd + Charbonnier + AdamW kernels)", "value": world * a.batch / dt, "n_gpus": world, "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": host_ms,
                          "batch_per_gpu": a.batch, "arch": a.arch, "img": a.img, "dtype": a.dtype, "loss_scale": ls, "loss": float(loss),
                          "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "ranks": world,
                          "rank_devices": devices, "hip_graph": bool(a.graph),
                          "gradient_exchange": "none (1 GPU)" if world == 1 else f"RCCL all-reduce, {len(sink.buckets)} buckets overlapped with the reverse sweep"}))
    if a.kernels_json:                   # every rank runs the instrumented step (it contains the gradient exchange); rank 0 writes its report
        import bench
        rows = bench.kernel_breakdown(None, None, 1, fn=step)
    if a.kernels_json and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.kernels_json)), exist_ok=True)
        with open(a.kernels_json, "w") as f:
            json.dump(rows, f, indent=0)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
