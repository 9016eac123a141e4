#!/bin/bash
# round 5, run 3: what bounds attn_block at C >= 256?  Timing ablations (UF_ABL builds under ab/) of one LeWin block per deep-stage shape + phase stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
python scripts/r05_ablate.py "default"
for v in 6 7 8 9 4 5; do UFORMER_HIP_LIB=$R/ab/abl$v/libuformer_hip.so UF_ALLOW_OLDER_LIB=1 python scripts/r05_ablate.py "UF_ABL=$v"; done
python scripts/r05_ablate.py "default (again)"
python scripts/ubench.py stamps
} 2>&1 | grep -v Warning | tee $O/r05_run3_ablate.txt
