#!/bin/bash
# round 4, GPU call 12: linear_wgrad4 (256 x 256 tiles, 4-deep DMA ring) against the third version; dgelu GEMM epilogue with its operand requested before the main loop
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms')"; }
{
echo "== pytest wgrad / linear"; timeout 600 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_ops.py -m gpu -q -k "wgrad or linear or gemm" 2>&1 | tail -5
echo "== wgrad microbench, third version"; UF_WGRAD_V4=0 timeout 300 python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "x(256|512) |wgrad'"
echo "== wgrad microbench, fourth version"; timeout 300 python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "x(256|512) |wgrad'"
echo "== gemm microbench, dgelu operand requested in the epilogue"; UFORMER_HIP_LIB=$R/ab/auxlate/libuformer_hip.so timeout 300 python scripts/ubench_train.py gemm 2>/dev/null | grep -E "dc_mul|gemm'"
echo "== gemm microbench, dgelu operand requested before the main loop"; timeout 300 python scripts/ubench_train.py gemm 2>/dev/null | grep -E "dc_mul|gemm'"
for r in 1 2; do echo "train wgrad3 run $r: $(UF_WGRAD_V4=0 tb)"; echo "train wgrad4 run $r: $(tb)"; echo "train wgrad4 + aux late run $r: $(UFORMER_HIP_LIB=$R/ab/auxlate/libuformer_hip.so tb)"; done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run12.txt
