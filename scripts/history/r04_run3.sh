#!/bin/bash
# round 4, GPU call 3: LDS-staged stem / head, XCD-contiguous Downsample tiles (tests + A/B), census of attn_block (true shader clock)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
echo "== tests"; python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
for r in 1 2; do echo "v1 stem/head run $r: $(UF_INPUT_PROJ_V1=1 UF_OUTPUT_PROJ_V1=1 b --kernels-json $O/k_sh1.json)"; echo "v2 stem/head run $r: $(b --kernels-json $O/k_sh2.json)"; done
for v in sh1 sh2; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json | tail -3; python - <<PY
import json
for r in json.load(open("$O/k_$v.json")):
    if r["kernel"].startswith(("gemm", "input_proj", "output_proj")): print("   %-44s %7.1f us" % (r["kernel"], r["ms_per_launch"] * 1e3))
PY
done
echo "== census attn"; python scripts/ubench.py census2 2>/dev/null
echo "== default bench line (summary block)"; python bench.py --no-cpu-baseline --train-steps 2 2>/dev/null | tail -c 1800
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run3.txt
