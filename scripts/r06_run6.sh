#!/bin/bash
# round 6, run 6: GELU and GELU' of the fused depthwise backward from one exp2 / rcp pair (gelu_and_grad_t): parity + same-box A/B against the two-call build
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -x -q 2>&1 | tail -5) | tee $O/r06_run6_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step  host enqueue', round(d['host_enqueue_ms_per_step'],1), 'ms  peak', round(d['peak_mem_gb'],1), 'GB')" "$1"; }
for i in 1 2 3; do
  UFORMER_HIP_LIB=$PWD/ab/nopair/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "two calls (exp2, rcp twice) #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "one exp2 / rcp pair        #$i"
done | tee $O/r06_run6_ab.txt
(python scripts/ubench_train.py stencil; UFORMER_HIP_LIB=$PWD/ab/nopair/libuformer_hip.so python scripts/ubench_train.py stencil) 2>/dev/null | grep -i "bwd\|total" | tee $O/r06_run6_dw.txt
