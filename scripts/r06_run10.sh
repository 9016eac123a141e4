#!/bin/bash
# round 6, run 10: LayerNorm backward with one partial row per workgroup and 2048 workgroups (8 waves per SIMD) against the 512-workgroup form (ab/oldln)
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bwd.py -m gpu -x -q -k "layernorm or lewin_block or model_backward or uformer_B" 2>&1 | tail -4) | tee $O/r06_run10_pytest.txt
cat > /tmp/lnb.py <<'P'
import sys, torch
sys.path.insert(0, ".")
from uformer_amd import ops
B = 32
for (H, C) in ((256, 32), (256, 64), (128, 64), (128, 128), (64, 128), (64, 256), (32, 256), (32, 512), (16, 512)):
    M = B * H * H
    x = torch.randn(M, C, device="cuda"); g = torch.randn(C, device="cuda"); dy = torch.randn(M, C, device="cuda").to(torch.bfloat16); add = torch.randn(M, C, device="cuda")
    f = lambda: ops.layernorm_bwd_fused(x, g, dy, B, H, H, add=add, cast=dict(scale=None, windowed=True, shift=4))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"layernorm_bwd_cast {H}x{H}x{C}: {us:8.1f} us  {M * C * 16 / us / 1e6:7.0f} GB/s")
P
echo "--- new (2048 workgroups, one partial row each)" | tee $O/r06_run10_ln.txt; python /tmp/lnb.py 2>/dev/null | tee -a $O/r06_run10_ln.txt
echo "--- old (512 workgroups x RPB partial rows)" | tee -a $O/r06_run10_ln.txt; UFORMER_HIP_LIB=$PWD/ab/oldln/libuformer_hip.so python /tmp/lnb.py 2>/dev/null | tee -a $O/r06_run10_ln.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  UFORMER_HIP_LIB=$PWD/ab/oldln/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "LN bwd 512 workgroups  #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "LN bwd 2048 workgroups #$i"
done | tee $O/r06_run10_ab.txt
