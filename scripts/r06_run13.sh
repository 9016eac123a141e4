#!/bin/bash
# round 6, run 13: weight-gradient chunk counts in whole multiples of 8 (the same number of chunks on every XCD): kernel table + parity + A/B of the training step against ab/oldwg
O=gpurun_out; mkdir -p $O
(echo "=== new (chunks a multiple of 8)"; python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "^wgrad|\{"; echo "=== old"; UFORMER_HIP_LIB=$PWD/ab/oldwg/libuformer_hip.so python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "^wgrad|\{") | tee $O/r06_run13_wgrad.txt | grep -E "===|qkv|\{"
(timeout 900 python -m pytest tests/test_gpu_bwd.py -m gpu -x -q -k "wgrad or lewin_block or model_backward or uformer_B or uformer_T" 2>&1 | tail -3) | tee $O/r06_run13_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  UFORMER_HIP_LIB=$PWD/ab/oldwg/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "chunks as they came   #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "chunks multiple of 8  #$i"
done | tee $O/r06_run13_ab.txt
