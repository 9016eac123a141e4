#!/bin/bash
# round 4, GPU call 5: full suite on the current defaults; stem with weights from global memory vs first form; determinism over 6 fresh processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
echo "== pytest -m gpu"; python -m pytest tests -m gpu -q 2>&1 | tail -8
for i in 1 2 3 4 5 6; do echo "fresh process $i: $(python scripts/out_hash.py 2>/dev/null)"; done
for r in 1 2; do echo "stem v1 run $r: $(UF_INPUT_PROJ_V1=1 b --kernels-json $O/k_s1.json)"; echo "stem v2 run $r: $(b --kernels-json $O/k_s2.json)"; done
for v in s1 s2; do python - <<PY
import json
for r in json.load(open("$O/k_$v.json")):
    if r["kernel"].startswith(("input_proj", "output_proj")): print("$v   %-44s %7.1f us" % (r["kernel"], r["ms_per_launch"] * 1e3))
PY
done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run5.txt
