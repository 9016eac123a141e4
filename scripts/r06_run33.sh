#!/bin/bash
# round 6, run 33: the LDS-staged head with 1 / 2 (shipped) / 4 pixel columns per thread (tile 16 / 32 / 64 pixels wide: 28 / 52 / 102 KB of LDS = 5 / 3 / 1 workgroups per CU)
O=gpurun_out; mkdir -p $O
python - <<'PY' 2>/dev/null | tee $O/r06_run33_head.txt
import os, subprocess, sys
code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from uformer_amd import ops, packing
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import hashlib
for B in (8, 16, 32):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B * 256 * 256, 64, device="cuda", generator=g)
    w = packing.pack_output_proj(torch.randn(3, 64, 3, 3, device="cuda", generator=g) / 24.0)
    b = torch.randn(3, device="cuda", generator=g); img = torch.rand(B, 3, 256, 256, device="cuda", generator=g)
    y = ops.output_proj(x, w, b, B, 256, 256, img); torch.cuda.synchronize()
    print(f"output_proj batch {B}: {timeit(lambda: ops.output_proj(x, w, b, B, 256, 256, img)):8.1f} us  sha {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]}")
"""
for name, lib in (("NX=2 (shipped)", None), ("NX=1", "ab/opnx1/libuformer_hip.so"), ("NX=4", "ab/opnx4/libuformer_hip.so"), ("NX=2 (shipped)", None), ("NX=1", "ab/opnx1/libuformer_hip.so")):
    env = dict(os.environ)
    if lib: env["UFORMER_HIP_LIB"] = os.path.join(os.getcwd(), lib)
    print("===", name, flush=True)
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout, flush=True)
PY
