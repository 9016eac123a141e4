#!/bin/bash
# round 6, run 8: why is the f16 mode 3 % slower than bf16?  per-kernel HIP-event tables of both operand types on one box (UF_STREAMS=1 inside the instrumented forwards)
O=gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-other-modes"
for i in 1 2; do
$B --dtype bf16 --kernels-json $O/r06_run8_k_bf16.json 2>/dev/null | python scripts/print_bench.py "bf16 #$i"
$B --dtype f16 --kernels-json $O/r06_run8_k_f16.json 2>/dev/null | python scripts/print_bench.py "f16  #$i"
done | grep -v train | tee $O/r06_run8_ab.txt
python - <<'P' | tee -a $O/r06_run8_ab.txt
import json
a = {r["kernel"].replace("bf16", "T"): r for r in json.load(open("gpurun_out/r06_run8_k_bf16.json"))}
b = {r["kernel"].replace("f16", "T"): r for r in json.load(open("gpurun_out/r06_run8_k_f16.json"))}
tot = [0, 0]
for k in sorted(a, key=lambda k: -a[k]["ms"]):
    if k in b:
        ma, mb = a[k]["ms"] / 3, b[k]["ms"] / 3
        tot[0] += ma; tot[1] += mb
        if ma > 0.03:
            print(f"{k:48s} bf16 {ma:7.3f} ms/step  f16 {mb:7.3f}  ({100 * (mb / ma - 1):+5.1f} %)")
print("sum of matched kernels: bf16 %.3f  f16 %.3f ms/step" % tuple(tot))
P
