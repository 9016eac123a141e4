#!/usr/bin/env python3
"""Micro-benchmark single C-ABI entry points on the Uformer-B stage shapes (MI355X)."""
import os
import sys
import time

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
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "lngemm"
    dt = torch.bfloat16
    if what == "lngemm":
        for (M, C, heads, H) in ((65536, 256, 8, 64), (16384, 512, 16, 32), (65536, 128, 4, 64), (1048576, 64, 2, 256)):
            B = M // (H * H)
            x = torch.randn(M, C, device="cuda")
            g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            wq = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(dt)
            bq = torch.zeros(3 * C, device="cuda")
            w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(dt)
            b1 = torch.zeros(4 * C, device="cuda")
            t_q = timeit(lambda: ops.ln_qkv(x, g, b, wq, bq, heads, B=B, H=H, W=H, shift=4))
            t_f = timeit(lambda: ops.ln_linear_gelu(x, g, b, w1, b1))
            print(f"dbg={os.environ.get('UF_LNGEMM_DBG', '0')} M={M} C={C}: ln_qkv {t_q:7.1f} us ({2 * M * 3 * C * C / t_q / 1e6:6.1f} TF/s)   "
                  f"ln_fc1 {t_f:7.1f} us ({2 * M * 4 * C * C / t_f / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
