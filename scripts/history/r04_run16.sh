#!/bin/bash
# round 4, GPU call 16: stale-row diagnosis 7 (scripts/r04_dbg7.py) and the first-form stem with 8 pixels per thread
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bi() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p --kernels-json $O/k_$1.json 2>/dev/null | python scripts/print_bench.py "$2"; }
{
timeout 600 python scripts/r04_dbg7.py 2>&1 | grep -v "^$"
echo "== pytest stem / model"; timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -k "input_proj or stem or determinism or fresh or golden" 2>&1 | tail -3
for r in 1 2; do UF_INPUT_PROJ_PX=4 bi px4 "stem 4 px/thread run $r"; UF_INPUT_PROJ_PX=8 bi px8 "stem 8 px/thread run $r"; done
python - <<'P'
import json
for t in ("px4","px8"):
    d=json.load(open(f"gpurun_out/k_{t}.json")); ks=d if isinstance(d,list) else d.get("kernels",d)
    for k in ks:
        if "input_proj" in k["kernel"]: print(t, k["kernel"], round(1e3*k.get("ms_per_launch", k.get("ms",0)/max(1,k.get("launches",1))),1), "us")
P
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run16.txt
