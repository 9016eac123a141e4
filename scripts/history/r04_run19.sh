#!/bin/bash
# round 4, GPU call 19: scripts/ubench_hip/pk_opsel -- every packed-f32 op_sel form of the library's kernels beside an idle SIMD, MFMA waves, LDS reads
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ timeout 300 ./scripts/ubench_hip/pk_opsel; } 2>&1 | tee $O/r04_run19.txt
