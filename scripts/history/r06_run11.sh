#!/bin/bash
# round 6, run 11: dgamma / dbeta sums of the LayerNorm backward on the side stream: parity + A/B of the training step
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -x -q 2>&1 | tail -4) | tee $O/r06_run11_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --ln-sums-inline 2>/dev/null | show "LN sums on the critical stream #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "LN sums on the side stream     #$i"
done | tee $O/r06_run11_ab.txt
