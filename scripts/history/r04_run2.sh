#!/bin/bash
# round 4, GPU call 2: full GPU suite on the new defaults (UF_MCONV 1, LDS-DMA GEMM staging, linear_wgrad3), GEMM / weight-gradient
# microbenchmarks and the training step with the DMA paths on / off, leff2 with 8 producer waves, role stamps incl. consumers, census (clock).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms')"; }
{
echo "== pytest -m gpu"; python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== gemm microbench, DMA off"; UF_GEMM_DMA=0 python scripts/ubench_train.py gemm 2>/dev/null | grep -E "64x64x256|32x32x512|128x128x128|\{"
echo "== gemm microbench, DMA on"; python scripts/ubench_train.py gemm 2>/dev/null | grep -E "64x64x256|32x32x512|128x128x128|\{"
echo "== wgrad microbench, v2"; UF_WGRAD_DMA=0 python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "64x64x256|32x32x512|128x128x128|256x256x64|\{"
echo "== wgrad microbench, v3"; python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "64x64x256|32x32x512|128x128x128|256x256x64|\{"
for r in 1 2; do
echo "train both off run $r: $(UF_GEMM_DMA=0 UF_WGRAD_DMA=0 tb)"
echo "train gemm dma run $r: $(UF_WGRAD_DMA=0 tb)"
echo "train wgrad3   run $r: $(UF_GEMM_DMA=0 tb)"
echo "train both on  run $r: $(tb)"
done
for r in 1 2; do echo "infer default run $r: $(b --kernels-json $O/k_def.json)"; echo "infer pw8 run $r: $(UF_LEFF2_VARIANT=p b --kernels-json $O/k_pw8.json)"; done
for v in def pw8; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json; done
echo "== stamps2 default"; python scripts/ubench.py stamps2 2>/dev/null
echo "== stamps2 pw8"; UF_LEFF2_VARIANT=p python scripts/ubench.py stamps2 2>/dev/null
echo "== census"; python scripts/ubench.py census 2>/dev/null
echo "== stamps attn"; python scripts/ubench.py stamps 2>/dev/null
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run2.txt
