#!/usr/bin/env python3
"""Round 4, packed-operand-select hazard, exposure of attn_block: a LeWin block (C = 32 and C = 128) on a side stream while the main stream runs the MFMA GEMM,
compared bit for bit with the same block run alone -- for the build whose LayerNorm sums contain `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`
(UFORMER_HIP_LIB=ab/attn_slp/libuformer_hip.so) and for the shipped build (attn_block without SLP vectorisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uformer_amd import model as um, ops

torch.manual_seed(0)
side = torch.cuda.Stream()
ga = torch.randn(131072, 256, device="cuda").to(torch.bfloat16); gw = torch.randn(1024, 256, device="cuda").to(torch.bfloat16); gb = torch.zeros(1024, device="cuda")
for (B, H, C, heads) in ((8, 256, 32, 1), (8, 128, 128, 4)):
    blk = um.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
    x = torch.randn(B, H * H, C, device="cuda")
    with torch.no_grad():
        ref = blk(x, None, torch.bfloat16)
        torch.cuda.synchronize()
        bad, nel = 0, 0
        for rep in range(60):
            for _ in range(4):
                ops.linear(ga, gw, gb)
            with torch.cuda.stream(side):
                y = blk(x, None, torch.bfloat16)
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
                nel += int((y != ref).sum())
    print(f"{os.environ.get('TAG', '')}: LeWin block C={C} {H}x{H} B={B} on a side stream beside the GEMM: {bad} of 60 outputs differ from the block run alone ({nel} elements)", flush=True)
