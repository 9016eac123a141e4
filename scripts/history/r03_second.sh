#!/bin/bash
# round 3, second GPU call: full GPU suite, prefetch-distance sweeps (bit-identity checked), per-kernel tables of the best settings
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/r03_pytest_gpu.txt
for f in parity_grad_B_f32 parity_grad_B_bf16 parity_grad_B_f16 parity_grad_T_f32 parity_grad_T_bf16 parity_grad_T_f16 parity_model parity_ops; do cp $O/$f.json $O/r03_$f.json 2>/dev/null; done
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; }
{
echo "hash base   $(python scripts/out_hash.py 2>/dev/null)"
echo "hash pf     $(UF_PF_ATTN=1024 UF_PF_LEFF2=768 python scripts/out_hash.py 2>/dev/null)"
for pa in 0 256 512 1024 2048; do echo "UF_PF_ATTN=$pa UF_PF_LEFF2=0 : $(UF_PF_ATTN=$pa b)"; done
for pl in 256 768 1536 3072; do echo "UF_PF_ATTN=0 UF_PF_LEFF2=$pl : $(UF_PF_LEFF2=$pl b)"; done
for c in "512 768" "1024 768" "1024 1536" "2048 1536"; do set -- $c; echo "UF_PF_ATTN=$1 UF_PF_LEFF2=$2 : $(UF_PF_ATTN=$1 UF_PF_LEFF2=$2 b)"; done
echo "repeat base : $(b)"
} | tee $O/r03_prefetch_sweep.txt
UF_STREAMS=1 python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --kernels-json $O/r03_k_base.json > /dev/null 2>&1
UF_STREAMS=1 UF_PF_ATTN=1024 UF_PF_LEFF2=768 python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --kernels-json $O/r03_k_pf.json > /dev/null 2>&1
python scripts/kernel_table.py $O/r03_k_base.json | tee $O/r03_k_base.txt; python scripts/kernel_table.py $O/r03_k_pf.json | tee $O/r03_k_pf.txt
