#!/usr/bin/env python3
"""scripts/parity_one.py [dtype ...]: max-abs error of Uformer-B 256x256 (one image, synthetic trained-like weights) against the CPU oracle,
for the library UFORMER_HIP_LIB selects.  Test infrastructure: the oracle is the checker here, never the thing measured."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import uformer_oracle as O
from uformer_amd import model as um, spec

cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234); x = spec.synth_input(1, 256, 256, 1234)
torch.set_num_threads(16)
ref = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
for name in (sys.argv[1:] or ["f16", "bf16"]):
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[name]
    m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=dt).eval()
    m.load_state_dict(sd); m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda()).float().cpu()
    d = (y - ref)
    print(f"{os.environ.get('UFORMER_HIP_LIB', 'default')} {name}: max abs err vs oracle {d.abs().max().item():.3e}  mean abs {d.abs().mean().item():.3e}  PSNR {O.psnr(y, ref):.1f} dB")
