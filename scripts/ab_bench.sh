#!/bin/bash
# scripts/ab_bench.sh <rounds> <variant> [<variant> ...]   ("base" = the in-tree library)
# interleaved bench runs on ONE box, so box-to-box variance (a few %) cancels out of the comparison
R=${GRAFT_REPO_ROOT:-$(pwd)}; rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernels-json $R/gpurun_out/ab_${v}_$r.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $r, round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', ' gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"
  done
done
