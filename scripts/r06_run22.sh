#!/bin/bash
# round 6, run 22: PipelinedForward test (all depths), deeper rings at small batch, and the bench line with modes.pipelined
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "pipelined or graph or stream" 2>&1 | tail -3) | tee $O/r06_run22_pytest.txt
python - <<'PY' 2>/dev/null | tee $O/r06_run22_depth.txt
import os, sys, time, statistics, json
sys.path.insert(0, os.getcwd())
import torch
from uformer_amd import infer, model, spec
dev = torch.device("cuda:0")
cfg = spec.arch_config("Uformer_B", img_size=256)
m = model.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator, dd_in=cfg.dd_in, compute_dtype=torch.bfloat16)
m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
m = m.to(dev).eval()
with torch.no_grad():
    for B in (1, 2, 4, 8, 16):
        x = spec.synth_input(B, 256, 256, 7).to(dev)
        n = max(40, 320 // B)
        row = {"batch": B}
        for depth in (1, 2, 3, 4, 6):
            pf = infer.PipelinedForward(m, depth=depth)
            def loop():
                for _ in pf.map(x for _ in range(n)):
                    pass
            loop()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row[f"depth{depth}_img_s"] = round(B * n / statistics.median(ts), 1)
        print(json.dumps(row))
PY
python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p 2>$O/r06_run22_bench.err | tail -1 > $O/r06_run22_bench.json; python scripts/print_bench.py $O/r06_run22_bench.json 2>/dev/null | tail -30; tail -c 1500 $O/r06_run22_bench.json
