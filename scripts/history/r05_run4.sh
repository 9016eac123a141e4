#!/bin/bash
# round 5, run 4: non-temporal hints on attn_block's activation streams (UF_NT builds), small-batch regime eager vs HIP-graph replay
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
python scripts/r05_ablate.py "default"
for v in 1 3 7; do UFORMER_HIP_LIB=$R/ab/nt$v/libuformer_hip.so UF_ALLOW_OLDER_LIB=1 python scripts/r05_ablate.py "UF_NT=$v"; done
python scripts/r05_ablate.py "default (again)"
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_run4_nt.txt
python scripts/r05_smallbatch.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_run4_smallbatch.txt
