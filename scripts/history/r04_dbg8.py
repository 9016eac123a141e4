#!/usr/bin/env python3
"""Round 4, stale-row diagnosis 8: the LDS-staged stem ALONE on a side stream while the main stream runs one kind of kernel, compared element by element
with the first form: which co-runner makes it go wrong, and where (image, row, column, channel) the wrong elements sit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uformer_amd import model as um, ops

torch.manual_seed(0)
B, H, E = 8, 256, 32
img = torch.rand(B, 3, H, H, device="cuda")
w27 = (torch.randn(27, E, device="cuda") * 0.2).contiguous()
bias = (torch.randn(E, device="cuda") * 0.1).contiguous()
os.environ["UF_INPUT_PROJ_V2"] = "0"
ref = ops.input_proj(img, w27, bias)
os.environ["UF_INPUT_PROJ_V2"] = "1"
alone = ops.input_proj(img, w27, bias)
print("alone, same stream: equal", torch.equal(ref, alone), flush=True)

side = torch.cuda.Stream()
C = 32
h1 = torch.randn(B, H, H, 4 * C, device="cuda").to(torch.bfloat16)
w9 = torch.randn(9, 4 * C, device="cuda") * 0.2; bd = torch.randn(4 * C, device="cuda") * 0.1
w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16); b2 = torch.randn(C, device="cuda")
xs = torch.randn(B * H * H, C, device="cuda")
blk = um.LeWinTransformerBlock(C, (H, H), 1, win_size=8, shift_size=0, modulator=True).cuda().eval()
xb = torch.randn(B, H * H, C, device="cuda")
big = torch.randn(64 * 1024 * 1024, device="cuda")
ga = torch.randn(131072, 256, device="cuda").to(torch.bfloat16); gw = torch.randn(1024, 256, device="cuda").to(torch.bfloat16); gb = torch.zeros(1024, device="cuda")
img2 = torch.rand(B, 3, H, H, device="cuda")


def blk_fwd():
    with torch.no_grad():
        blk(xb, None, torch.bfloat16)


loads = {
    "nothing": lambda: None,
    "torch elementwise (mul_)": lambda: big.mul_(1.0),
    "GEMM (uf_linear_fwd 131072x1024x256)": lambda: ops.linear(ga, gw, gb),
    "leff2 (LDS-DMA)": lambda: ops.dwconv_linear2(h1, w9, bd, w2, b2, xs),
    "LeWin block C=32 (attn_block + leff2)": blk_fwd,
    "the LDS-staged stem itself": lambda: ops.input_proj(img2, w27, bias),
    "first-form stem": None,
}
for name, fn in loads.items():
    bad_runs, where = 0, None
    for rep in range(30):
        if name == "first-form stem":
            os.environ["UF_INPUT_PROJ_V2"] = "0"
            for _ in range(2):
                ops.input_proj(img2, w27, bias)
            os.environ["UF_INPUT_PROJ_V2"] = "1"
        else:
            for _ in range(3):
                fn()
        with torch.cuda.stream(side):
            y = ops.input_proj(img, w27, bias)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad_runs += 1
            if where is None:
                d = (y != ref).reshape(B, H, H, E)
                idx = d.nonzero()
                where = dict(n=int(d.sum()), images=sorted(set(idx[:, 0].tolist())), rows=sorted(set(idx[:, 1].tolist()))[:12], cols=sorted(set(idx[:, 2].tolist()))[:12],
                             chans=sorted(set(idx[:, 3].tolist())), maxdiff=float((y - ref).abs().max()),
                             row_mod4=sorted(set((idx[:, 1] % 4).tolist())), col_mod64=sorted(set((idx[:, 2] % 64).tolist()))[:16])
    print(f"co-runner {name}: {bad_runs} of 30 side-stream stem outputs differ from the first form", where if where else "", flush=True)
