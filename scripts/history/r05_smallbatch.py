#!/usr/bin/env python3
"""Small-batch regime (VERDICT r04 "next" 8; the reference's eval scripts run batch 1, test/test_sidd.py:101-107): Uformer-B 256x256 bf16 at batch
1 / 2 / 4 / 8 / 16, eager forward vs HIP-graph replay (uformer_amd.infer.GraphedForward), img/s and ms per forward."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uformer_amd import infer, model as um, spec

cfg = spec.arch_config("Uformer_B", img_size=256)
sd = spec.synth_state_dict(cfg, 1234)
m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
               dd_in=cfg.dd_in, compute_dtype=torch.bfloat16).eval()
m.load_state_dict(sd, strict=True)
m = m.cuda()


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {}
with torch.no_grad():
    for B in (1, 2, 4, 8, 16):
        x = spec.synth_input(B, 256, 256, 77 + B).cuda()
        n = 60 if B <= 4 else 30
        te = timeit(lambda: m(x), n)
        gf = infer.GraphedForward(m, x)
        same = bool(torch.equal(gf(x), m(x)))
        tg = timeit(lambda: gf(x), n)
        out[B] = {"eager_ms": 1e3 * te, "eager_img_s": B / te, "graph_ms": 1e3 * tg, "graph_img_s": B / tg, "bit_identical": same}
        print(f"batch {B:2d}: eager {1e3 * te:7.3f} ms = {B / te:7.1f} img/s | graph replay {1e3 * tg:7.3f} ms = {B / tg:7.1f} img/s | bit-identical {same}", flush=True)
print("JSON " + json.dumps(out))
