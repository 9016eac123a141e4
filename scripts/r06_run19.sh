#!/bin/bash
# round 6, run 19: (a) forward stencil: XCD groups of 1 / 2 / 4 / 8 / 16 strips against plain launch order (0) and an eighth of the tensor per XCD (-1);
# (b) attention backward: two neighbouring heads (one 128-byte line of a dO row) on one XCD against head-fastest launch order; (c) the training step
O=gpurun_out; mkdir -p $O
for v in 0 1 2 4 8 16 -1; do echo "=== dwxcd=$v"; UF_VARIANT="dwxcd=$v" python scripts/ubench_train.py stencil 2>/dev/null | grep -E "^dwconv_pre_gelu|\{"; done | tee $O/r06_run19_dw.txt | grep -E "===|\{"
for v in 0 1 0 1; do echo "=== attpair=$v"; UF_VARIANT="attpair=$v" python scripts/ubench_train.py attn 2>/dev/null | grep -E "^attn|\{"; done | tee $O/r06_run19_attn.txt
(timeout 1200 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_ops.py -m gpu -x -q -k "dwconv or stencil or attention or attn or lewin_block" 2>&1 | tail -3) | tee $O/r06_run19_pytest.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  UF_VARIANT="dwxcd=0,attpair=0" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "dwxcd=0 attpair=0    #$i"
  UF_VARIANT="dwxcd=0,attpair=1" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "dwxcd=0 attpair=1    #$i"
  UF_VARIANT="dwxcd=4,attpair=1" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "dwxcd=4 attpair=1    #$i"
  UF_VARIANT="dwxcd=16,attpair=1" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "dwxcd=16 attpair=1   #$i"
done | tee $O/r06_run19_ab.txt
