#!/bin/bash
# round 4, GPU call 14: the operand copy written by the LayerNorm backward (uf_layernorm_bwd_cast), one pack launch per block, vectorised DropPath
# draws: tests, the step with each switch off, and where the remaining small ATen kernels of a step come from
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms', 'host', round(d.get('host_enqueue_ms_per_step',0),1))"; }
{
echo "== pytest test_gpu_bwd + tail"; timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_tail.py -m gpu -q -x 2>&1 | tail -6
for r in 1 2; do
echo "train run $r: $(tb)"
echo "train, separate grad_fork passes run $r: $(UF_LN_BWD_CAST=0 tb)"
echo "train, five pack launches per block run $r: $(UF_PACK_LAUNCHES=5 tb)"
done
echo "train f16: $(tb --dtype f16)"
timeout 300 python scripts/r04_fills.py 2>&1 | tail -60
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run14.txt
