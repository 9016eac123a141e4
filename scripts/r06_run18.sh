#!/bin/bash
# round 6, run 18: XCD-aware position order in the two depthwise walking kernels (forward: blockIdx -> the XCD's eighth of the positions; backward: slot groups): stencil table,
# parity, and the training step against ab/base (the library before the change)
O=gpurun_out; mkdir -p $O
(echo "=== new (XCD-aware)"; python scripts/ubench_train.py stencil 2>/dev/null | grep -E "^dwconv_pre_gelu|^dwconv_bwd_fused|\{"; echo "=== base"; UFORMER_HIP_LIB=$PWD/ab/base/libuformer_hip.so python scripts/ubench_train.py stencil 2>/dev/null | grep -E "^dwconv_pre_gelu|^dwconv_bwd_fused|\{") | tee $O/r06_run18_dw.txt
(timeout 1200 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_ops.py -m gpu -x -q -k "dwconv or stencil or leff or lewin_block or model_backward or uformer_B or uformer_T" 2>&1 | tail -3) | tee $O/r06_run18_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  UFORMER_HIP_LIB=$PWD/ab/base/libuformer_hip.so python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "base                 #$i"
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "XCD-aware stencils   #$i"
done | tee $O/r06_run18_ab.txt
