#!/usr/bin/env python3
"""Throughput of Uformer-B 256 x 256 (bf16, batch 16 per forward) with 1 / 2 / 3 / 4 forwards of successive batches in flight (uformer_amd.infer.PipelinedForward)
against the eager loop; bit-identity of the outputs.  UF_STREAMS (read once per process) sets how many parts ONE forward is cut into.
    python scripts/pipelined_bench.py [--batch 16] [--steps 40]"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uformer_amd import infer, model, spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = spec.arch_config("Uformer_B", img_size=256)
    sd = spec.synth_state_dict(cfg, 1234)
    T = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    m = model.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator, dd_in=cfg.dd_in,
                      compute_dtype=T)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    xs = [spec.synth_input(a.batch, 256, 256, 1234 + i).to(dev) for i in range(4)]
    with torch.no_grad():
        ref = [m(x).clone() for x in xs]
    out = {"batch": a.batch, "steps": a.steps, "dtype": a.dtype, "UF_STREAMS": os.environ.get("UF_STREAMS", "default")}

    def region(fn):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / a.steps)
        return statistics.median(ts)

    with torch.no_grad():
        def eager():
            for i in range(a.steps):
                m(xs[i & 3])
        eager()
        dt = region(eager)
        out["eager_ms"] = dt * 1e3
        out["eager_img_s"] = a.batch / dt
        for depth in (1, 2, 3, 4):
            pf = infer.PipelinedForward(m, depth=depth)
            ys = list(pf.map(xs))
            torch.cuda.synchronize()
            same = all(torch.equal(y, r) for y, r in zip(ys, ref))

            def piped():
                for _ in pf.map(xs[i & 3] for i in range(a.steps)):
                    pass
            piped()
            dt = region(piped)
            out[f"depth{depth}_ms"] = dt * 1e3
            out[f"depth{depth}_img_s"] = a.batch / dt
            out[f"depth{depth}_bit_identical"] = same
    print(json.dumps(out))


if __name__ == "__main__":
    main()
