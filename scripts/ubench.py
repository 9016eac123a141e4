#!/usr/bin/env python3
"""Micro-benchmark single C-ABI entry points on the Uformer-B stage shapes (MI355X)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

# uf_debug_set_tbuf buffer: sampled phase stamps in the first 65536 entries, then the per-workgroup census
# (8 entries per workgroup, uf_common.h struct Census) for up to 32768 workgroups
TBUF_ELEMS = 65536 + 8 * 32768

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


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] == "lngemm"):
    main()


def bench_blocks():
    """attention half / LeFF half of one block on the big stages, with the ablation env vars."""
    import ctypes
    from uformer_amd import _lib, model, packing
    lib = _lib.load()
    for (B, H, C, heads) in ((16, 64, 256, 8), (16, 32, 512, 16), (16, 64, 128, 4), (16, 256, 64, 2)):
        blk = model.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
        bp = blk._pack(torch.bfloat16)
        M = B * H * H
        x = torch.randn(M, C, device="cuda")
        nbytes = lib.uf_block_workspace_bytes(M, C, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        ta = timeit(lambda: lib.uf_lewin_attn_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st))
        tl = timeit(lambda: lib.uf_leff_fwd(bp, x.data_ptr(), C, B, H, H, C, 1, ws.data_ptr(), nbytes, st))
        print(f"A={os.environ.get('UF_ATTNBLK_DBG', '0')} L={os.environ.get('UF_LEFF2_DBG', '0')} M={M} C={C}: attn_half {ta:7.1f} us   leff_half {tl:7.1f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "blocks":
    bench_blocks()


def bench_stamps():
    """Phase timestamps of attn_block (cycle counter of wave 0..N of every 64th window)."""
    from uformer_amd import _lib, model
    lib = _lib.load()
    for (B, H, C, heads) in ((16, 64, 256, 8), (16, 32, 512, 16), (16, 64, 128, 4), (16, 128, 128, 4), (16, 32, 256, 8)):
        blk = model.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
        bp = blk._pack(torch.bfloat16)
        M = B * H * H
        x = torch.randn(M, C, device="cuda")
        nbytes = lib.uf_block_workspace_bytes(M, C, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        waves = 8 if (C == 512 or (C == 256 and M // 64 <= 256)) else 4
        tb = torch.zeros(TBUF_ELEMS, dtype=torch.int64, device="cuda")
        for _ in range(3):
            lib.uf_lewin_block_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        lib.uf_lewin_block_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        t = tb[:65536].cpu().reshape(-1, waves, 16)
        t0 = t[..., 0].min()
        print(f"C={C}: stamps relative to first block start, wave 0 of sampled blocks (cycles @100MHz? raw counter units)")
        for bi in range(min(t.shape[0], 6)):
            row = t[bi, 0]
            print(f"  block {bi * 64:5d}: start {int(row[0] - t0):8d} | LN {int(row[1] - row[0]):7d} bar {int(row[2] - row[1]):6d} QKV {int(row[3] - row[2]):7d} "
                  f"[pack+S {int(row[8] - row[3]):6d} softmax {int(row[9] - row[8]):6d} PV {int(row[10] - row[9]):6d} store {int(row[4] - row[10]):6d}] rest-units {int(row[5] - row[4]):7d} bar {int(row[6] - row[5]):6d} proj {int(row[7] - row[6]):7d} resid+LN2 {int(row[11] - row[7]):6d} fc1 {int(row[12] - row[11]):7d} | total {int(row[12] - row[0]):7d}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stamps":
    bench_stamps()


def bench_stamps_leff2():
    from uformer_amd import _lib, model
    lib = _lib.load()
    for (B, H, C, heads) in ((16, 64, 256, 8), (16, 32, 512, 16), (16, 64, 128, 4), (16, 256, 64, 2)):
        blk = model.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
        bp = blk._pack(torch.bfloat16)
        M = B * H * H
        x = torch.randn(M, C, device="cuda")
        nbytes = lib.uf_block_workspace_bytes(M, C, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        nblk = M // 64
        tb = torch.zeros(TBUF_ELEMS, dtype=torch.int64, device="cuda")
        for _ in range(3):
            lib.uf_leff_fwd(bp, x.data_ptr(), C, B, H, H, C, 1, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        lib.uf_leff_fwd(bp, x.data_ptr(), C, B, H, H, C, 1, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        t = tb[:65536].cpu().reshape(-1, 16, 4).float()[:4]
        n = 4 * C // 64 + 1
        act = t[..., 0] > 0
        pr = act & (t[..., 2] > 0)
        co = act & (t[..., 2] == 0)
        print(f"leff2 C={C}: per iteration (cycles), producers work {t[..., 0][pr].mean() / n:.0f} wait {t[..., 1][pr].mean() / n:.0f} | "
              f"consumers work {t[..., 0][co].mean() / n:.0f} wait {t[..., 1][co].mean() / n:.0f}   ({n} iterations, {int(pr.sum()) // 4} producer waves)")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stamps2":
    bench_stamps_leff2()


def bench_stamps_lngemm():
    from uformer_amd import _lib
    lib = _lib.load()
    dt = torch.bfloat16
    for (M, C) in ((65536, 256), (16384, 512), (65536, 128), (1048576, 64)):
        x = torch.randn(M, C, device="cuda")
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(dt)
        b1 = torch.zeros(4 * C, device="cuda")
        for _ in range(3):
            ops.ln_linear_gelu(x, g, b, w1, b1)
        tb = torch.zeros(TBUF_ELEMS, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        ops.ln_linear_gelu(x, g, b, w1, b1)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        t = tb[:65536].cpu().reshape(-1, 4, 4).float()
        t = t[t[:, 0, 3] > 0][:4]
        nu = t[:, :, 3].mean()
        print(f"ln_fc1 M={M} C={C}: per wave (cycles): LN phase {t[:, :, 0].mean():.0f} | k-loops total {t[:, :, 1].mean():.0f} epilogues total {t[:, :, 2].mean():.0f} "
              f"| units/block {nu:.0f} -> per unit: kloop {t[:, :, 1].mean() / (nu / 4):.0f} epilogue {t[:, :, 2].mean() / (nu / 4):.0f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stamps3":
    bench_stamps_lngemm()


def census_report(name, tb, nblocks, wall_us=None):
    """Decode the per-workgroup census the kernels write at tbuf[65536 + 8 b] (uf_common.h, struct Census)."""
    import collections
    c = tb[65536:65536 + nblocks * 8].cpu().reshape(nblocks, 8).numpy()
    c = c[c[:, 1] > 0]
    cyc = (c[:, 1] - c[:, 0]).astype(float)
    rt = (c[:, 3] - c[:, 2]).astype(float) * 10.0          # ns (100 MHz counter)
    hw = c[:, 4]
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xf) + ((hw >> 12) & 1) * 16 + ((hw >> 13) & 7) * 32   # xcc, cu, sh, se
    span_ns = (c[:, 3].max() - c[:, 2].min()) * 10.0
    per_cu = collections.defaultdict(list)
    for k, a, b in zip(cu_key, c[:, 2], c[:, 3]):
        per_cu[int(k)].append((int(a), int(b)))
    maxov = []
    for k, iv in per_cu.items():
        ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv])
        cur = best = 0
        for _, d in ev:
            cur += d
            best = max(best, cur)
        maxov.append(best)
    import numpy as np
    print(f"{name}: {len(c)} workgroups on {len(per_cu)} CUs, workgroups/CU {len(c) / len(per_cu):.2f}, max concurrent per CU: "
          f"mean {np.mean(maxov):.2f} max {max(maxov)} | block life {rt.mean() / 1e3:.1f} us = {cyc.mean():.0f} cycles -> clock {cyc.sum() / rt.sum():.2f} GHz"
          f" | kernel span {span_ns / 1e3:.1f} us")


def bench_census():
    from uformer_amd import _lib
    lib = _lib.load()
    dt = torch.bfloat16
    for (B, H, C) in ((16, 64, 256), (16, 32, 512), (16, 64, 128), (16, 128, 128)):
        M = B * H * H
        x = torch.randn(M, C, device="cuda")
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(dt)
        b1 = torch.zeros(4 * C, device="cuda")
        w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(dt)
        b2 = torch.zeros(C, device="cuda")
        wd = torch.randn(9, 4 * C, device="cuda") * 0.2
        bd = torch.zeros(4 * C, device="cuda")
        tb = torch.zeros(TBUF_ELEMS, dtype=torch.int64, device="cuda")
        for _ in range(2):
            h1 = ops.ln_linear_gelu(x, g, b, w1, b1)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        h1 = ops.ln_linear_gelu(x, g, b, w1, b1)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        census_report(f"ln_fc1 M={M} C={C}", tb, 8192)
        xr = x.clone()
        for _ in range(2):
            ops.dwconv_linear2(h1.reshape(B, H, H, 4 * C), wd, bd, w2, b2, xr)
        tb.zero_()
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        ops.dwconv_linear2(h1.reshape(B, H, H, 4 * C), wd, bd, w2, b2, xr)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        census_report(f"leff2  M={M} C={C}", tb, 8192)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "census":
    bench_census()


def bench_census_attn():
    """Census of the fused attention-half kernel on the stage shapes: workgroups per CU, block life in ns (100 MHz wall clock) and in
    s_memtime cycles -> the shader clock the kernel really ran at."""
    from uformer_amd import _lib, model
    lib = _lib.load()
    for (B, H, C, heads) in ((16, 64, 256, 8), (16, 32, 512, 16), (16, 32, 256, 8), (16, 64, 128, 4), (16, 128, 128, 4), (16, 256, 64, 2), (16, 128, 64, 2), (16, 256, 32, 1)):
        blk = model.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
        bp = blk._pack(torch.bfloat16)
        M = B * H * H
        x = torch.randn(M, C, device="cuda")
        nbytes = lib.uf_block_workspace_bytes(M, C, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        tb = torch.zeros(TBUF_ELEMS, dtype=torch.int64, device="cuda")
        for _ in range(3):
            lib.uf_lewin_attn_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
        t_us = timeit(lambda: lib.uf_lewin_attn_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st), n=10, warm=1)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(tb.data_ptr())
        lib.uf_lewin_attn_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        lib.uf_debug_set_tbuf(None)
        census_report(f"attn_block M={M} C={C} ({t_us:.1f} us)", tb, min(M // 64, 32768))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "census2":
    bench_census_attn()
