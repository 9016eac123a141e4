#!/bin/bash
# round 6, run 5: one ordered-sum launch per block for the four weight gradients (uf_linear_wgrad_partial + uf_column_sums): parity + A/B of the training step
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -x -q 2>&1 | tail -5) | tee $O/r06_run5_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step  host enqueue', round(d['host_enqueue_ms_per_step'],1), 'ms  peak', round(d['peak_mem_gb'],1), 'GB')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --no-defer-sums 2>/dev/null | show "per-call sums  #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "one sum / block #$i"
done | tee $O/r06_run5_ab.txt
