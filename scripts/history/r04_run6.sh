#!/bin/bash
# round 4, GPU call 6: persistent leff2 (tile walk with a continuous DMA ring): tests, determinism under poisoned workspaces, A/B against one tile per workgroup
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
echo "== tests"; python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_tail.py -m gpu -q 2>&1 | tail -6
echo "== poisoned workspace, default"; python scripts/r04_dbg5.py 2>&1 | grep -v amdgpu.ids
for r in 1 2; do echo "one tile per workgroup run $r: $(UF_LEFF2_PERSIST=0 b --kernels-json $O/k_np.json)"; echo "persistent run $r: $(b --kernels-json $O/k_p.json)"; done
for v in np p; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json; done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run6.txt
