#!/bin/bash
# round 4, GPU call 18: diagnosis 8 again with two builds of the LDS-staged stem: every pixel value moved into a register of its own before the packed FMAs
# (no op_sel high-half operand reads), and a long wait + gap between the LDS reads and their first use
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "== default build"; timeout 300 python scripts/r04_dbg8.py 2>&1 | grep -v "^$" | cut -c1-330
echo "== pixel values in registers of their own (UF_IP2_DBG=130)"; UFORMER_HIP_LIB=$R/ab/ip130/libuformer_hip.so timeout 300 python scripts/r04_dbg8.py 2>&1 | grep -v "^$" | cut -c1-330
echo "== wait + 32 idle cycles behind the LDS reads (UF_IP2_DBG=258)"; UFORMER_HIP_LIB=$R/ab/ip258/libuformer_hip.so timeout 300 python scripts/r04_dbg8.py 2>&1 | grep -v "^$" | cut -c1-330
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run18.txt
