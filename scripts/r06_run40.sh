#!/bin/bash
# round 6, run 40: the input gradient of Downsample from an LDS patch of dy, now rounding every tap's product to T and adding the taps in col2im's order (bit-identical to the patch-matrix route)
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bwd.py -m gpu -q -k "downsample_input_gradient" 2>&1 | tail -12) | tee $O/r06_run40_pytest.txt
for i in 1 2; do echo "=== patch form"; python scripts/ubench_down_bwd.py 2>/dev/null; echo "=== patch-matrix route"; UF_VARIANT="downdx=1" python scripts/ubench_down_bwd.py 2>/dev/null; done | tee $O/r06_run40_bwd.txt
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -q -k "uformer_B or uformer_T or traj or model_backward or tiny32" 2>&1 | tail -3) | tee -a $O/r06_run40_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step', d.get('loss'))" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "dx from an LDS patch   #$i"
  UF_VARIANT="downdx=1" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "patch-matrix route     #$i"
done | tee $O/r06_run40_ab.txt
