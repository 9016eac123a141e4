#!/usr/bin/env python3
"""HBM traffic per launch of every library kernel, from the FETCH_SIZE / WRITE_SIZE PMC passes
(scripts/pmc_passes.sh -> <dir>/pmcC_pmc.csv, <dir>/pmcD_pmc.csv), keyed by the library's timing symbol so
bench.py can attach it to `roofline.traffic`.

    python scripts/pmc_traffic.py gpurun_out profiles/r02_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts
128-B requests of wide (16 B/lane) streaming reads as 64 B, so it is DOUBLED here -- every hot read in these kernels
is a 16-B/lane load.  WRITE_SIZE is taken as reported (uncalibrated in the guide)."""
import collections
import csv
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]


def symbol(kernel):
    m = re.search(r"attn_block_kernel<uf::bf16, (\d+), (\d+)(?:, \d+)*>", kernel)                   # <T, C, NT[, LR[, TR]]>
    if m:
        return f"attn_block_fc1_bf16_c{m.group(1)}_nt{m.group(2)}"
    m = re.search(r"leff2_kernel<uf::bf16, (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, \d+)?>", kernel)      # <T, C, NPG, NC, NBUF, WPS[, PW[, CP]]>
    if m:
        return f"leff2_bf16_c{m.group(1)}_np{int(m.group(6) or 4) * int(m.group(2))}_nc{m.group(3)}"
    m = re.search(r"gemm_kernel<uf::bf16, (\d+), \d+, \d+, (\d+), (\d+)(?:, (true|false))?>", kernel)     # <T, BN, WGM, WGN, AL, EP[, DMA]>
    if m:
        return f"gemm_bf16_bn{m.group(1)}_a{m.group(2)}_e{m.group(3)}" + ("_dma" if m.group(4) == "true" else "")
    if "input_proj_kernel" in kernel or "input_proj2_kernel" in kernel:
        return "input_proj"
    if "output_proj_kernel" in kernel or "output_proj2_kernel" in kernel:
        return "output_proj"
    return None


def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        s = symbol(r["kernel"])
        if s:
            n = int(r["dispatches"])
            acc[s][0] += float(r[counter]) * 1024.0 * n
            acc[s][1] += n
    return {s: (b / n, n) for s, (b, n) in acc.items()}


import os
def first(*names):
    for n in names:
        if os.path.exists(f"{src}/{n}"):
            return f"{src}/{n}"
    raise SystemExit(f"none of {names} under {src}")
fetch, write = load(first("pmcC_pmc.csv", "r04_final_pmcC.csv", "r03_final_pmcC.csv", "r02_final_pmcC.csv"), "FETCH_SIZE"), load(first("pmcD_pmc.csv", "r04_final_pmcD.csv", "r03_final_pmcD.csv", "r02_final_pmcD.csv"), "WRITE_SIZE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha: the stamp bench.py checks before it quotes these numbers)
def busy_fractions():
    """Per symbol, from the SQ passes A (SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU) and B (SQ_VALU_MFMA_BUSY_CYCLES) when they are there: the fraction of the
    SIMD-cycles of a launch in which the MFMA pipe / the VALU was busy.  Normalisation (checked on attn_block<256,256>: 116.6 us = 245 K cycles): SQ_BUSY_CYCLES sums
    32 units that tick for the length of the launch (7.99 M = 32 x 250 K) while the MFMA / VALU counters sum over the 1024 SIMDs, i.e. over 32 x as many
    SIMD-cycles as one busy unit has; SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_ACTIVE_INST_VALU quad-cycles (MI355X_MICROARCH.md)."""
    try:
        pa, pb = first("pmcA_pmc.csv", "r04_final_pmcA.csv"), first("pmcB_pmc.csv", "r04_final_pmcB.csv")
    except SystemExit:
        return {}
    busy = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for r in csv.DictReader(open(pa)):
        s = symbol(r["kernel"])
        if s and "SQ_BUSY_CYCLES" in r:
            n = int(r["dispatches"])
            busy[s][0] += float(r["SQ_BUSY_CYCLES"]) * n
            busy[s][2] += float(r.get("SQ_ACTIVE_INST_VALU", 0.0)) * n
    for r in csv.DictReader(open(pb)):
        s = symbol(r["kernel"])
        if s and "SQ_VALU_MFMA_BUSY_CYCLES" in r:
            busy[s][1] += float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) * int(r["dispatches"])
    return {s: {"mfma_busy_frac": m / (32.0 * b), "valu_active_frac": 4.0 * v / (32.0 * b)} for s, (b, m, v) in busy.items() if b > 0}


frac = busy_fractions()
out = {}
for s in sorted(fetch):
    f, n = fetch[s]
    w = write.get(s, (0.0, 0))[0]
    out[s] = {"fetch_bytes_per_launch": 2.0 * f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": 2.0 * f + w, "dispatches_profiled": n}
    out[s].update(frac.get(s, {}))
json.dump({"kernel_source_sha": bench.kernel_source_sha(), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-modes --no-train-mode --no-720p (UF_STREAMS=1); "
                     "FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; average over the launches of a symbol",
           "kernels": out}, open(dst, "w"), indent=1)
for s, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
    print(f"{s:40s} fetch {v['fetch_bytes_per_launch'] / 1e6:8.1f} MB  write {v['write_bytes_per_launch'] / 1e6:8.1f} MB  (n={v['dispatches_profiled']})"
          + (f"  MFMA pipe busy {v['mfma_busy_frac']:.3f}  VALU active {v['valu_active_frac']:.3f}" if "mfma_busy_frac" in v else ""))
