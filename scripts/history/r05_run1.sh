#!/bin/bash
# round 5, run 1: full GPU suite on the new tree + A/B of the halo-recompute LeFF (UF_LEFF3=0: the round-4 pair attn_block(+fc1) -> leff2)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) | tee $O/r05_run1_pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-train-mode --no-720p"
for i in 1 2; do
  UF_LEFF3=0 $B --kernels-json $O/r05_run1_k_leff3off.json 2>/dev/null | python scripts/print_bench.py "leff3=0 #$i"
  $B --kernels-json $O/r05_run1_k_leff3on.json 2>/dev/null | python scripts/print_bench.py "leff3=1 #$i"
done | tee $O/r05_run1_ab.txt
python - <<'PY' | tee -a $O/r05_run1_ab.txt
import json
for tag in ("leff3off", "leff3on"):
    rows = json.load(open(f"gpurun_out/r05_run1_k_{tag}.json"))
    print(tag, "total ms/3 steps", round(sum(r["ms"] for r in rows), 3))
    for r in rows[:24]:
        print(f"  {r['kernel']:<58} {r['launches']:4d} x {1e3 * r['ms_per_launch']:8.1f} us  = {r['ms']:7.3f} ms")
PY
