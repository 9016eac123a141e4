#!/usr/bin/env python3
"""Pretty-print a bench.py --kernels-json dump: per kernel/shape and per Uformer stage."""
import json
import sys

rows = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tot = sum(r["ms"] for r in rows) / steps
print(f"total GPU ms/step {tot:.3f}")
STAGE = {(1048576, 32): "enc0", (262144, 64): "enc1", (65536, 128): "enc2", (16384, 256): "enc3", (4096, 512): "bott",
         (16384, 512): "dec0", (65536, 256): "dec1", (262144, 128): "dec2", (1048576, 64): "dec3"}
per_stage, per_kind = {}, {}
for r in rows:
    name = r["kernel"]
    sym, _, shape = name.partition(" ")
    dims = [int(v) for v in shape.split("x")] if shape else []
    C = None
    M = dims[0] if dims else None
    kind = sym
    if sym.startswith("gemm"):
        e = sym.rsplit("_e", 1)[1]
        kind = {"1": "fc1", "2": "qkv", "3": "proj", "4": "fc2", "5": "down", "6": "up"}[e]
        Mm, N, K = dims
        C = {"fc1": K, "qkv": K, "proj": K, "fc2": N}.get(kind)
    elif sym.startswith("ln_gemm"):
        kind = "ln_qkv" if "_qkv_" in sym else "ln_fc1"
        C = dims[2]
    elif sym.startswith("leff2"):
        kind = "leff2"
        C = dims[1]
    elif sym.startswith("attn_block"):
        kind = "attn_block"
        C = dims[1]
    elif sym.startswith("window_attn"):
        M, C = dims[0] * 64, dims[1] * dims[2]
    elif sym.startswith("layernorm"):
        C = dims[1]
    elif sym.startswith("dwconv"):
        C = dims[1] // 4
    st = STAGE.get((M, C), "other")
    per_stage.setdefault(st, {}).setdefault(kind, 0.0)
    per_stage[st][kind] += r["ms"] / steps
    per_kind[kind] = per_kind.get(kind, 0.0) + r["ms"] / steps
kinds = sorted(per_kind, key=lambda k: -per_kind[k])
print(f"{'stage':6s}" + "".join(f"{k[:10]:>11s}" for k in kinds) + f"{'total':>9s}")
for st in ["enc0", "enc1", "enc2", "enc3", "bott", "dec0", "dec1", "dec2", "dec3", "other"]:
    d = per_stage.get(st, {})
    print(f"{st:6s}" + "".join(f"{d.get(k, 0):11.3f}" for k in kinds) + f"{sum(d.values()):9.3f}")
print(f"{'all':6s}" + "".join(f"{per_kind[k]:11.3f}" for k in kinds) + f"{tot:9.3f}")
if "-v" in sys.argv:
    for r in sorted(rows, key=lambda r: -r["ms"]):
        ms = r["ms"] / r["launches"]
        print(f"{r['kernel']:48s} n={r['launches'] // steps:3d} {ms * 1e3:8.1f} us {r['flops'] / r['ms'] / 1e9:7.1f} TF/s "
              f"{r['bytes'] / r['ms'] / 1e6:8.1f} GB/s  {r['ms'] / steps:6.3f} ms/step")
