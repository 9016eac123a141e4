#!/usr/bin/env python3
"""Times the building blocks of the training backward on the Uformer-B stage shapes at batch 32 (MI355X):
   python scripts/ubench_train.py [stencil|wgrad|attn|all]   -> one line per (op, shape): microseconds, effective GB/s or TFLOP/s."""
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
    return e0.elapsed_time(e1) / n * 1e3  # us


# (H, C, heads) of the nine stages; batch 32
STAGES = [(256, 32, 1), (128, 64, 2), (64, 128, 4), (32, 256, 8), (16, 512, 16), (32, 512, 16), (64, 256, 8), (128, 128, 4), (256, 64, 2)]
UNIQUE = [(256, 32, 1), (256, 64, 2), (128, 64, 2), (128, 128, 4), (64, 128, 4), (64, 256, 8), (32, 256, 8), (32, 512, 16), (16, 512, 16)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B, T = 32, torch.bfloat16
    tot = {}
    for (H, C, heads) in UNIQUE:
        M, C4 = B * H * H, 4 * C
        tag = f"{H}x{H}x{C}"
        if what in ("stencil", "all"):
            h = torch.randn(B, H, H, C4, device="cuda").to(T)
            a = torch.randn(B, H, H, C4, device="cuda").to(T)
            w9, bias = torch.randn(9, C4, device="cuda") * 0.3, torch.randn(C4, device="cuda") * 0.1
            nbytes = h.numel() * 2
            for name, fn, passes in (("dwconv_plain", lambda: ops.dwconv3x3(h, w9, bias, gelu=False), 2), ("dwconv_pre_gelu", lambda: ops.dwconv3x3_pre_gelu(h, w9, bias), 3),
                                     ("dwconv_mul_dgelu", lambda: ops.dwconv3x3_mul_dgelu(h, w9, a), 3), ("dwconv_wgrad", lambda: ops.dwconv3x3_wgrad(h, a), 2),
                                     ("dwconv_bwd_fused", lambda: ops.dwconv3x3_bwd(h, w9, a), 3),
                                     ("gelu_fwd", lambda: ops.gelu(h), 2)):
                us = timeit(fn)
                tot[name] = tot.get(name, 0) + us
                print(f"{name:18s} {tag:14s} {us:9.1f} us  {passes * nbytes / us / 1e3:7.0f} GB/s")
            del h, a
        if what in ("wgrad", "all"):
            for (N, K, nm) in ((C4, C, "lin1"), (C, C4, "lin2"), (3 * C, C, "qkv"), (C, C, "proj")):
                dy = torch.randn(M, N, device="cuda").to(T)
                x = torch.randn(M, K, device="cuda").to(T)
                us = timeit(lambda: ops.linear_wgrad(dy, x))
                tot["wgrad"] = tot.get("wgrad", 0) + us
                print(f"wgrad_{nm:5s}        {tag:14s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {(M * (N + K) * 2) / us / 1e3:7.0f} GB/s")
                del dy, x
        if what in ("gemm", "all"):
            # the GEMMs of one block in the op-by-op training step: forward linear1 (two outputs), input gradients dc (times GELU'), dz, dO, dxn
            z = torch.randn(M, C, device="cuda").to(T)
            a4 = torch.randn(M, C4, device="cuda").to(T)
            q3 = torch.randn(M, 3 * C, device="cuda").to(T)
            w1, w1t = (torch.randn(C4, C, device="cuda") / C ** 0.5).to(T), (torch.randn(C, C4, device="cuda") / C ** 0.5).to(T)
            wq, wqt = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(T), (torch.randn(C, 3 * C, device="cuda") / C ** 0.5).to(T)
            wp = (torch.randn(C, C, device="cuda") / C ** 0.5).to(T)
            b4, b1, zero4, zero1 = torch.randn(C4, device="cuda"), torch.randn(C, device="cuda"), torch.zeros(C4, device="cuda"), torch.zeros(C, device="cuda")
            x32 = torch.randn(M, C, device="cuda")
            for name, fn, n, k, by in (("fc1_pre_gelu", lambda: ops.linear_pre_gelu(z, w1, b4), C4, C, M * (C + 2 * C4) * 2),
                                       ("dc_mul_dgelu", lambda: ops.linear_mul_dgelu(z, w1, zero4, a4), C4, C, M * (C + 2 * C4) * 2),
                                       ("dz (N=C,K=4C)", lambda: ops.linear(a4, w1t, zero1), C, C4, M * (C4 + C) * 2),
                                       ("fc2_residual", lambda: ops.linear_residual(a4, w1t, b1, x32, None, B, H, H), C, C4, M * (C4 * 2 + C * 8)),
                                       ("dxn (N=C,K=3C)", lambda: ops.linear(q3, wqt, zero1), C, 3 * C, M * (3 * C + C) * 2),
                                       ("dO (N=C,K=C)", lambda: ops.linear(z, wp, zero1), C, C, M * 2 * C * 2),
                                       ("qkv", lambda: ops.qkv(z, wq, torch.zeros(3 * C, device="cuda"), heads), 3 * C, C, M * 4 * C * 2)):
                us = timeit(fn)
                tot["gemm"] = tot.get("gemm", 0) + us
                print(f"gemm {name:14s} {tag:14s} {us:9.1f} us  {2.0 * M * n * k / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s")
            del z, a4, q3, x32
        if what in ("attn", "all"):
            nW, hd = M // 64, 32
            q = (torch.randn(nW, heads, 64, hd, device="cuda") * hd ** -0.5).to(T)
            k = torch.randn(nW, heads, 64, hd, device="cuda").to(T)
            vt = torch.randn(nW, heads, hd, 64, device="cuda").to(T)
            bias = torch.randn(heads, 64, 64, device="cuda")
            do = torch.randn(M, C, device="cuda").to(T)
            us = timeit(lambda: ops.window_attention_bwd_qkv(q, k, vt, bias, do, H, H, 4))
            tot["attn_bwd"] = tot.get("attn_bwd", 0) + us
            print(f"attn_bwd_qkv       {tag:14s} {us:9.1f} us  {7.0 * M * C * 2 / us / 1e3:7.0f} GB/s")
    print({k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
