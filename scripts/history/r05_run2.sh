#!/bin/bash
# round 5, run 2: leff3 as a software pipeline (one barrier per slot): parity subset + A/B against the round-4 pair
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-run2}
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "leff or lewin or model or variants" 2>&1 | tail -5) | tee $O/r05_${TAG}_pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-train-mode --no-720p"
for i in 1 2; do
  UF_LEFF3=0 $B 2>/dev/null | python scripts/print_bench.py "leff3=0 #$i"
  $B --kernels-json $O/r05_${TAG}_k_leff3on.json 2>/dev/null | python scripts/print_bench.py "leff3=1 #$i"
done | tee $O/r05_${TAG}_ab.txt
python - $TAG <<'PY' | tee -a $O/r05_${TAG}_ab.txt
import json, sys
rows = json.load(open(f"gpurun_out/r05_{sys.argv[1]}_k_leff3on.json"))
print("total ms/3 steps", round(sum(r["ms"] for r in rows), 3))
for r in rows:
    if "leff3" in r["kernel"] or "c64" in r["kernel"] or "c32" in r["kernel"]:
        print(f"  {r['kernel']:<58} {r['launches']:4d} x {1e3 * r['ms_per_launch']:8.1f} us  = {r['ms']:7.3f} ms")
PY
