#!/usr/bin/env python3
"""uf_downsample_bwd (dW, db, dx of Conv2d k4 s2 p1 on tokens) at the four levels of Uformer-B 256 x 256, batch 32: microseconds per call.
    python scripts/ubench_down_bwd.py      (UF_VARIANT="downdx=1": the patch-matrix route for dx)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uformer_amd import ops


def timeit(fn, n=10, warm=2):
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
    B, T, tot = int(os.environ.get("BATCH", "32")), torch.bfloat16, 0.0
    g = torch.Generator(device="cuda").manual_seed(5)
    for (H, C) in ((256, 32), (128, 64), (64, 128), (32, 256)):
        x = torch.randn(B * H * H, C, device="cuda", generator=g)
        dy = torch.randn(B * H * H // 4, 2 * C, device="cuda", generator=g)
        wpt = (torch.randn(16 * C, 2 * C, device="cuda", generator=g) / (16 * C) ** 0.5).to(T)
        skip = torch.zeros(B * H * H, C, device="cuda")
        us = timeit(lambda: ops.downsample_bwd(x, dy, wpt, B, H, H, add_to=skip))
        tot += us
        print(f"downsample_bwd {H}x{H}x{C:<4d} {us:9.1f} us")
    print(f"total {tot:.0f} us")


if __name__ == "__main__":
    main()
