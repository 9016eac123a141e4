#!/bin/bash
# same-box A/B of the UF_HOIST variants (ab/h0 = off, in-tree = C<=128, ab/h3 = + C=256, ab/h7 = every width): bit-identity first, then interleaved bench rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for v in base h0 h3 h7; do if [ $v = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi; echo "hash $v $(python scripts/out_hash.py 2>/dev/null)"; done
for r in 1 2 3; do for v in h0 base h3 h7; do if [ $v = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi
  python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --kernels-json $O/k_$v.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v run $r', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; done; done
} | tee $O/r03_hoist_ab.txt
unset UFORMER_HIP_LIB
for v in h0 base h3 h7; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json | grep -E "stage|enc|bott|dec|all"; done | tee $O/r03_hoist_tables.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -3
