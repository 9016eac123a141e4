#!/bin/bash
# round 4, GPU call 17: stale-row diagnosis 8 (scripts/r04_dbg8.py): the LDS-staged stem alone on a side stream beside one kind of kernel at a time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ timeout 600 python scripts/r04_dbg8.py 2>&1 | grep -v "^$"; } 2>&1 | grep -v amdgpu.ids | tee $O/r04_run17.txt
