#!/bin/bash
# round 5, run 7: LLVM scheduling strategies for the two fused kernels (attn_block, leff2): per-stage kernel times of one LeWin block, all nine stage shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
python scripts/r05_ablate.py --all "default"
for v in s_ilp s_mem s_lat; do UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so UF_ALLOW_OLDER_LIB=1 python scripts/r05_ablate.py --all "$v"; done
python scripts/r05_ablate.py --all "default (again)"
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_run7_sched.txt
