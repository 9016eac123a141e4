#!/bin/bash
# round 4, GPU call 9: window_attn_bwd2 with the operand tiles prefetched a window ahead and the bias in LDS (3 waves per SIMD with 11 spilled dwords vs 2 without)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms')"; }
{
echo "== pytest test_gpu_bwd"; python -m pytest tests/test_gpu_bwd.py -m gpu -q 2>&1 | tail -4
echo "== attn bwd microbench, first version"; UF_ATTN_BWD_V1=1 python scripts/ubench_train.py attn 2>/dev/null | tail -1
echo "== attn bwd microbench, second version (3 waves / SIMD)"; python scripts/ubench_train.py attn 2>/dev/null
echo "== attn bwd microbench, second version (2 waves / SIMD)"; UFORMER_HIP_LIB=$R/ab/bwd2w2/libuformer_hip.so python scripts/ubench_train.py attn 2>/dev/null
for r in 1 2; do echo "train attn_bwd v1 run $r: $(UF_ATTN_BWD_V1=1 tb)"; echo "train attn_bwd v2 (3) run $r: $(tb)"; echo "train attn_bwd v2 (2) run $r: $(UFORMER_HIP_LIB=$R/ab/bwd2w2/libuformer_hip.so tb)"; done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run9.txt
