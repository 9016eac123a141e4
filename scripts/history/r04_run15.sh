#!/bin/bash
# round 4, GPU call 15: stencil that activates its input (linear1 stores one tensor less), 16-lane column sums, uf_leff_bwd's fused residual add
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms', 'peak', round(d.get('peak_mem_gb',0),1), 'GB')"; }
{
echo "== pytest test_gpu_bwd + tail"; timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_tail.py -m gpu -q 2>&1 | tail -6
for r in 1 2; do
echo "train run $r: $(tb)"
echo "train, linear1 stores both tensors run $r: $(UF_DW_GELU_IN=0 tb)"
echo "train, column sums with 1 / 4 lanes run $r: $(UF_COLSUM_LANES=1 tb)"
done
echo "train, recompute mode: $(UF_TRAIN_RECOMPUTE=1 tb)"
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run15.txt
