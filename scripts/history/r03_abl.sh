#!/bin/bash
# timing-only ablations of attn_block (outputs are wrong by construction): ab/abl4 = no h1 stores, ab/abl5 = no row stores in phase 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in base abl4 abl5; do if [ $v = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi
  UF_STREAMS=1 python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --kernels-json $O/k_$v.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'img/s  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"
  python scripts/kernel_table.py $O/k_$v.json | grep -E "stage|enc|bott|dec|all"; done | tee $O/r03_abl.txt
