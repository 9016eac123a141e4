#!/bin/bash
# round 5, run 8: depth of leff2's halo DMA ring (NBUF): is the kernel bound by HBM latency / (slots in flight)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
python scripts/r05_ablate.py --all "default"
for v in "$@"; do UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so UF_ALLOW_OLDER_LIB=1 python scripts/r05_ablate.py --all "$v"; done
python scripts/r05_ablate.py --all "default (again)"
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_run8_ring.txt
