#!/bin/bash
# round 6, run 34: the A-stationary form of dY W2 * GELU'(c) (uf_linear_mul_dgelu_fm) against the tiled GEMM: bit-identity, the stage shapes, the training step
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_abi_symbols.py -m gpu -q -k "a_stationary or epilogues or abi" 2>&1 | tail -5) | tee $O/r06_run34_pytest.txt
python scripts/ubench_train.py dc 2>/dev/null | grep -E "^dc|\{" | tee $O/r06_run34_dc.txt
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -q -x -k "lewin_block or uformer_B or uformer_T or traj or model_backward" 2>&1 | tail -3) | tee -a $O/r06_run34_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "A-stationary dc   #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --tiled-dc 2>/dev/null | show "tiled dc          #$i"
done | tee $O/r06_run34_ab.txt
