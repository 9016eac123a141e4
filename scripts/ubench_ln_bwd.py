import sys, torch
sys.path.insert(0, ".")
from uformer_amd import ops
B = 32
tot = 0
for (H, C, n) in ((256, 32, 2), (256, 64, 2), (128, 64, 4), (128, 128, 4), (64, 128, 16), (64, 256, 16), (32, 256, 16), (32, 512, 16), (16, 512, 4)):
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
    tot += us * n
print(f"layernorm_bwd_cast, launches of one step: {tot / 1e3:.3f} ms")
