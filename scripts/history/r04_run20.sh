#!/bin/bash
# round 4, GPU call 20: the library without the hazardous packed operand selects (LDS-staged stem with its pixel values in registers of their own, now the
# default; attn_block built with -fno-slp-vectorize): tests, whole-model hashes over refilled workspaces, A/B against attn_block with SLP
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bi() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p --kernels-json $O/k_$1.json 2>/dev/null | python scripts/print_bench.py "$2"; }
{
echo "== pytest ops + model"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -4
echo "== whole-model hashes, ten forwards over refilled workspaces each"; timeout 400 python scripts/r04_dbg7.py 2>&1 | grep -v "^$" | head -4
echo "== stem alone beside other kernels"; timeout 200 python scripts/r04_dbg8.py 2>&1 | grep -v "^$" | cut -c1-200
for r in 1 2 3; do bi noslp "attn_block without SLP (default) run $r"; UFORMER_HIP_LIB=$R/ab/attn_slp/libuformer_hip.so bi slp "attn_block with SLP run $r"; done
UF_INPUT_PROJ_V2=0 bi stemv1 "first-form stem"
python scripts/kernel_table.py $O/k_noslp.json
python scripts/kernel_table.py $O/k_slp.json | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run20.txt
