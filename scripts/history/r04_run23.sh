#!/bin/bash
# round 4, GPU call 23: does the packed-operand-select hazard show in attn_block as it was built before (LayerNorm sums as v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0])?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ TAG="attn_block built WITH SLP vectorisation (before)" UFORMER_HIP_LIB=$R/ab/attn_slp/libuformer_hip.so timeout 300 python scripts/r04_dbg9.py 2>&1 | grep -v "^$"
  TAG="shipped build (attn_block without SLP)" timeout 300 python scripts/r04_dbg9.py 2>&1 | grep -v "^$"; } 2>&1 | grep -v amdgpu.ids | tee $O/r04_run23.txt
