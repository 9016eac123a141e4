#!/bin/bash
# round 6, run 35: GELU' of the dY W2 * GELU'(c) store evaluated two values at a time (packed f32 arithmetic) against the scalar form (ab/dgscalar): bit-identity (hash), the stage shapes, the step
O=gpurun_out; mkdir -p $O
python - <<'PY' 2>/dev/null | tee $O/r06_run35_dgelu.txt
import os, subprocess, sys
code = r"""
import os, sys, torch, hashlib
sys.path.insert(0, os.getcwd())
from uformer_amd import ops
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0; h = hashlib.sha256()
for (H, C) in ((256, 32), (256, 64), (128, 64), (128, 128), (64, 128), (64, 256), (32, 256), (32, 512), (16, 512)):
    M = 32 * H * H; g = torch.Generator(device="cuda").manual_seed(C + H)
    dy = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(4 * C, C, device="cuda", generator=g) / C ** 0.5).to(torch.bfloat16)
    c = (torch.randn(M, 4 * C, device="cuda", generator=g) * 2).to(torch.bfloat16)
    z = torch.zeros(4 * C, device="cuda")
    y = ops.linear_mul_dgelu(dy, w, z, c); torch.cuda.synchronize()
    h.update(y.view(torch.int16).cpu().numpy().tobytes())
    us = timeit(lambda: ops.linear_mul_dgelu(dy, w, z, c)); tot += us
    print(f"dc {H}x{H}x{C:<4d} {us:8.1f} us  {M * (C * 2 + 16 * C) / us / 1e3:6.0f} GB/s")
    del dy, c, y
print(f"total {tot:.0f} us  sha {h.hexdigest()[:16]}")
"""
for name, lib in (("packed", None), ("scalar", "ab/dgscalar/libuformer_hip.so"), ("packed", None), ("scalar", "ab/dgscalar/libuformer_hip.so")):
    env = dict(os.environ)
    if lib: env["UFORMER_HIP_LIB"] = os.path.join(os.getcwd(), lib)
    print("===", name, flush=True)
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout, flush=True)
PY
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "packed GELU' in the dc store   #$i"
  UFORMER_HIP_LIB=$PWD/ab/dgscalar/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "scalar                         #$i"
done | tee $O/r06_run35_ab.txt
