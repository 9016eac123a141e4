#!/usr/bin/env python3
"""SHA-256 of the restored batch (Uformer-B 256x256, batch 16, every operand type): A/B builds / environment switches that must not change
a single bit print the same line."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uformer_amd import model as um, spec  # noqa: E402

cfg = spec.arch_config("Uformer_B", img_size=256)
sd = spec.synth_state_dict(cfg, 1234)
x = spec.synth_input(int(os.environ.get("HASH_BATCH", "16")), 256, 256, 1234).cuda()
out = []
for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                   dd_in=cfg.dd_in, compute_dtype=dt).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    with torch.no_grad():
        y = m(x)
    torch.cuda.synchronize()
    out.append(f"{name}:{hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]}")
print(" ".join(out))
