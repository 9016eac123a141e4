#!/usr/bin/env python3
"""Calibration only (not a product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the stage shapes the
prototypes in scripts/ubench_hip/gemm_dma.hip and wgrad_dma.hip are measured on.  bf16, f32 accumulate."""
import torch


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, what) in [(131072, 256, 1024, "dz dec1"), (131072, 1024, 256, "fc1 dec1"), (32768, 512, 2048, "dz dec0"), (32768, 2048, 512, "fc1 dec0"),
                        (524288, 128, 512, "dz dec2"), (131072, 256, 768, "dxn dec1")]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(a, w.t()))
    print(f"A W^T   {what:10s} {M:7d}x{N:5d}x{K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
for (M, N, K, what) in [(131072, 1024, 256, "lin1 dec1"), (32768, 2048, 512, "lin1 dec0"), (32768, 512, 2048, "lin2 dec0")]:
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(dy.t(), x))
    print(f"dY^T X  {what:10s} {M:7d}x{N:5d}x{K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
