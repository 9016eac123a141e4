#!/bin/bash
# round 6, run 7: per-step operand packs up front on a side stream: parity + A/B of the training step
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py tests/test_gpu_tail.py -m gpu -x -q 2>&1 | tail -5) | tee $O/r06_run7_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step  host enqueue', round(d['host_enqueue_ms_per_step'],1), 'ms  peak', round(d['peak_mem_gb'],1), 'GB')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --no-prepack 2>/dev/null | show "packs in series  #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "packs up front   #$i"
done | tee $O/r06_run7_ab.txt
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 --dtype f16 2>/dev/null | show "f16 operands" | tee -a $O/r06_run7_ab.txt
