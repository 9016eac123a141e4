#!/bin/bash
# round 3, first GPU call: f16 parity tests, bench with every mode, image-chunk (Infinity Cache) A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/r03_pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|ok" | tail -5 | tee $O/r03_smoke.txt
cp $O/parity_model.json $O/r03_parity_model.json 2>/dev/null; cp $O/parity_ops.json $O/r03_parity_ops.json 2>/dev/null
for f in parity_grad_B_f32 parity_grad_B_bf16 parity_grad_B_f16; do cp $O/$f.json $O/r03_$f.json 2>/dev/null; done
python bench.py --kernels-json $O/r03_kernels_hip_events.json > $O/r03_bench_first.json 2> $O/r03_bench_first.err; cut -c1-1500 $O/r03_bench_first.json
for mb in 0 64 96 128 192; do for st in 1 2; do
  UF_CHUNK_MB=$mb UF_STREAMS=$st python bench.py --no-cpu-baseline --no-other-modes --no-train-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UF_CHUNK_MB=$mb UF_STREAMS=$st', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done | tee $O/r03_chunk_ab.txt
