#!/bin/bash
# round 6, run 20: workgroups a weight-gradient launch aims for (every chunk costs one N x K f32 partial written and read again): 256 / 384 / 512 (shipped) / 768 / 1024;
# two side streams for the weight-gradient jobs instead of one
O=gpurun_out; mkdir -p $O
for v in 256 384 512 768 1024; do echo "=== wgtarget=$v"; UF_VARIANT="wgtarget=$v" python scripts/ubench_train.py wgrad 2>/dev/null | grep -E "^wgrad|\{"; done | tee $O/r06_run20_wgrad.txt | grep -E "===|\{"
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2 3; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "shipped (512)        #$i"
  UF_VARIANT="wgtarget=256" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "wgtarget=256         #$i"
  UF_VARIANT="wgtarget=384" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "wgtarget=384         #$i"
  UF_VARIANT="wgtarget=768" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "wgtarget=768         #$i"
  UF_BWD_STREAMS=3 python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "two side streams     #$i"
done | tee $O/r06_run20_ab.txt
