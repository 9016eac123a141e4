#!/bin/bash
# round 4, GPU call 7: consumer waves take stencil jobs (UF_LEFF2_VARIANT=c): tests, A/B, determinism; bit-identity of the low-register attn_block forms
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
echo "== tests, variant c"; UF_LEFF2_VARIANT=c python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -4
echo "hash default:   $(python scripts/out_hash.py 2>/dev/null)"
echo "hash variant c: $(UF_LEFF2_VARIANT=c python scripts/out_hash.py 2>/dev/null)"
for v in 0 1 2; do echo "hash LR=$v:      $(UF_ATTN_LR=$v python scripts/out_hash.py 2>/dev/null)"; done
for r in 1 2; do echo "default run $r: $(b --kernels-json $O/k_d.json)"; echo "variant c run $r: $(UF_LEFF2_VARIANT=c b --kernels-json $O/k_c.json)"; done
for v in d c; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json; done
echo "== stamps2 default"; python scripts/ubench.py stamps2 2>/dev/null
echo "== stamps2 variant c"; UF_LEFF2_VARIANT=c python scripts/ubench.py stamps2 2>/dev/null
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run7.txt
