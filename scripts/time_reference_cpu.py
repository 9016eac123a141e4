#!/usr/bin/env python3
"""Time the REFERENCE ITSELF (/root/reference/model.py, imported unmodified behind the 3-symbol timm shim of
tests/golden/make_golden.py) on the CPU cores of the BUILD CONTAINER -- the GPU box has no /root/reference.  Uformer-B 256x256,
fp32, eval, no_grad, B = 1 and B = 4, median of 3 after a warm-up; written to profiles/r06_reference_cpu.json (round 2: r02_reference_cpu.json), which bench.py
carries as ``cpu_baseline.reference_container`` next to the oracle timing it measures on the GPU box's own cores.

    PYTHONDONTWRITEBYTECODE=1 python scripts/time_reference_cpu.py
"""
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_golden as mg  # noqa: E402
import make_golden_r2 as r2  # noqa: E402

spec = mg.spec


@torch.no_grad()
def main():
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    cfg = spec.arch_config("Uformer_B", img_size=256)
    m = r2.build_ref(cfg, spec.synth_state_dict(cfg, 1234))
    out = {"what": "reference model.py (ZhendongWang6/Uformer) Uformer-B 256x256 forward, fp32, eval, torch.no_grad, timed in the build container",
           "torch": torch.__version__, "cores": threads, "unit": "images/s"}
    for B in (1, 4):
        x = spec.synth_input(B, 256, 256, 1234)
        m(x)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); m(x); ts.append(time.perf_counter() - t0)
        out[f"b{B}_images_per_s"] = B / statistics.median(ts)
        out[f"b{B}_seconds_median_of_3"] = statistics.median(ts)
    out["value"] = max(out["b1_images_per_s"], out["b4_images_per_s"])
    out["round"] = 6
    path = os.path.join(REPO, "profiles", "r06_reference_cpu.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
