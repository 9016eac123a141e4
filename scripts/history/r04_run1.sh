#!/bin/bash
# round 4, GPU call 1: the MFMA depthwise stencil of leff2 (UF_MCONV 2 = default build, 1 = ab/mc1, 0 = ab/mc0 = the round-3 VALU stencil):
# correctness first (ops + model suites), then same-box A/B with per-stage tables, parity of the three forms, the VALU issue-rate
# microbenchmark and the role stamps of leff2.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
lib() { if [ $1 = mc2 ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$1/libuformer_hip.so; fi; }
{
echo "== tests (default = UF_MCONV 2)"; python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
echo "== tests (UF_MCONV 1)"; lib mc1; python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
for v in mc0 mc1 mc2; do lib $v; python scripts/parity_one.py f16 bf16 2>/dev/null; done
for r in 1 2; do for v in mc0 mc1 mc2; do lib $v; echo "$v run $r: $(b --kernels-json $O/k_$v.json)"; done; done
for v in mc0 mc2; do lib $v; echo "f16 $v: $(b --dtype f16)"; done
for v in mc0 mc1 mc2; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json; done
for v in mc0 mc2; do lib $v; echo "== stamps2 $v"; python scripts/ubench.py stamps2 2>/dev/null; python scripts/ubench.py blocks 2>/dev/null; done
unset UFORMER_HIP_LIB
echo "== valu_rate"; ./scripts/ubench_hip/valu_rate
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run1.txt
