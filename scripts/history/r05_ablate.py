#!/usr/bin/env python3
"""Per-kernel time of ONE LeWin block (uf_lewin_block_fwd: attn_block with its fc1 phase + leff2) at the deep-stage shapes of Uformer-B at batch 16,
from the library's own HIP-event instrumentation.  Run once per library build (UFORMER_HIP_LIB=ab/<variant>/libuformer_hip.so) to compare
timing ablations of attn_block (UF_ABL, uf_attnblk.hip)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uformer_amd import _lib, model

lib = _lib.load()
shapes = [(16, 64, 256, 8, "dec1"), (16, 32, 512, 16, "dec0"), (16, 32, 256, 8, "enc3"), (16, 16, 512, 16, "bott"), (32, 32, 512, 16, "dec0@B32"), (16, 64, 128, 4, "enc2")]
if "--all" in sys.argv:            # every stage shape of Uformer-B 256x256 at batch 16
    sys.argv.remove("--all")
    shapes = [(16, 256, 32, 1, "enc0"), (16, 128, 64, 2, "enc1"), (16, 64, 128, 4, "enc2"), (16, 32, 256, 8, "enc3"), (16, 16, 512, 16, "bott"),
              (16, 32, 512, 16, "dec0"), (16, 64, 256, 8, "dec1"), (16, 128, 128, 4, "dec2"), (16, 256, 64, 2, "dec3")]
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
out = {}
for (B, H, C, heads, name) in shapes:
    torch.manual_seed(C)
    blk = model.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
    bp = blk._pack(torch.bfloat16)
    M = B * H * H
    x = torch.randn(M, C, device="cuda")
    nbytes = lib.uf_block_workspace_bytes(M, C, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        lib.uf_lewin_block_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
    torch.cuda.synchronize()
    lib.uf_timing_enable(1)
    for _ in range(30):
        lib.uf_lewin_block_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
    torch.cuda.synchronize()
    lib.uf_timing_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.uf_timing_report(buf, len(buf))
    rows = json.loads(buf.value.decode())
    d = {r["kernel"].split(" ")[0]: 1e3 * r["ms"] / max(1, r["launches"]) for r in rows}
    a = [v for k, v in d.items() if k.startswith("attn_block")]
    l = [v for k, v in d.items() if k.startswith("leff")]
    out[name] = (a[0] if a else 0.0, l[0] if l else 0.0)
depth = {"enc0": 1, "enc1": 2, "enc2": 8, "enc3": 8, "bott": 2, "dec0": 8, "dec1": 8, "dec2": 2, "dec3": 1}
if all(n in depth for n in out):
    print(f"{tag:<22} all 40 blocks of a forward: {sum(depth[n] * (a + l) for n, (a, l) in out.items()) / 1e3:.3f} ms of kernel time")
print(f"{tag:<22}" + "  ".join(f"{n}: attn {a:6.1f} leff {l:6.1f}" for n, (a, l) in out.items()), flush=True)
