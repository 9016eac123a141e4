#!/bin/bash
# round 6, run 17: training-step A/B of two grid knobs against the shipped library: LayerNorm backward with at most 1024 workgroups (shipped: 2048), attention backward with 512 / heads chunks (shipped: 1024 / heads)
O=gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "shipped                      #$i"
  UFORMER_HIP_LIB=$PWD/ab/ln1024/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "LN bwd <= 1024 workgroups    #$i"
  UFORMER_HIP_LIB=$PWD/ab/attg512/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "attention bwd 512/heads      #$i"
done | tee $O/r06_run17_ab.txt
