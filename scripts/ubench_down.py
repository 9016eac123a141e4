#!/usr/bin/env python3
"""Downsample (Conv2d k4 s2 p1 as an implicit GEMM on the token layout) at the four levels of Uformer-B 256 x 256: microseconds, TFLOP/s, GB/s of compulsory
traffic; a hash of the outputs (bit-identity across library builds).   python scripts/ubench_down.py [--batch 16] [--dtype bf16]"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uformer_amd import ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    T = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    B, tot, h = a.batch, 0.0, hashlib.sha256()
    g = torch.Generator(device="cuda").manual_seed(7)
    for (H, C) in ((256, 32), (128, 64), (64, 128), (32, 256)):
        x = torch.randn(B * H * H, C, device="cuda", generator=g)
        w = (torch.randn(2 * C, 16 * C, device="cuda", generator=g) / (16 * C) ** 0.5).to(T)
        b = torch.randn(2 * C, device="cuda", generator=g)
        wf = ops.pack_weight_fm(w) if (T != torch.float32 and os.environ.get("DOWN_NO_FM") is None) else None      # DOWN_NO_FM=1: the row-major weight only
        y = ops.downsample(x, w, b, B, H, H, w_fm=wf)
        torch.cuda.synchronize()
        h.update(y.cpu().numpy().tobytes())
        us = timeit(lambda: ops.downsample(x, w, b, B, H, H, w_fm=wf))
        M, N, K = B * H * H // 4, 2 * C, 16 * C
        by = x.numel() * 4 + y.numel() * 4 + w.numel() * w.element_size()
        tot += us
        print(f"downsample {H}x{H}x{C:<4d} M={M:<7d} N={N:<4d} K={K:<5d} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s")
    print(f"total {tot:.1f} us   outputs sha256 {h.hexdigest()[:16]}")


if __name__ == "__main__":
    main()
