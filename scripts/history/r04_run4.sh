#!/bin/bash
# round 4, GPU call 4: flakiness check of the batch-independence test, low-register attn_block (UF_ATTN_LR 0 / 1 / 2): tests + A/B,
# shader clock / throttling experiment (scripts/r04_clock.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
for i in 1 2 3 4 5 6; do echo "batch_independence run $i: $(python -m pytest tests/test_gpu_model.py -m gpu -x -q -k 'batch_independence or concurrent' 2>&1 | tail -1)"; done
echo "== tests UF_ATTN_LR=1"; UF_ATTN_LR=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
echo "== out_hash: LR 0 / 1 / 2"; for v in 0 1 2; do echo "LR=$v $(UF_ATTN_LR=$v python scripts/out_hash.py 2>/dev/null)"; done
for r in 1 2; do for v in 0 1 2; do echo "LR=$v run $r: $(UF_ATTN_LR=$v b --kernels-json $O/k_lr$v.json)"; done; done
for v in 0 1 2; do echo "== LR=$v"; python scripts/kernel_table.py $O/k_lr$v.json | grep -E "stage|enc0|enc1|enc2|dec2|dec3|all"; done
echo "== clock experiment"; python scripts/r04_clock.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run4.txt
