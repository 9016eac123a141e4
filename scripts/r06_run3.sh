#!/bin/bash
# round 6, run 3: attn_block single-operand-tile form at C = 256 (UF_ATTN_ST=1: three workgroups per CU) against the two-tile form
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "variants or lewin_block" 2>&1 | tail -4) | tee $O/r06_run3_pytest.txt
B="python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-f32-mode"
for i in 1 2 3; do
  UF_ATTN_ST=0 $B 2>/dev/null | python scripts/print_bench.py "two tiles (ST=0) #$i"
  UF_ATTN_ST=1 $B 2>/dev/null | python scripts/print_bench.py "single tile (ST=1) #$i"
done | grep -v train | tee $O/r06_run3_ab.txt
for bs in 32 64; do
  UF_ATTN_ST=0 $B --batch $bs 2>/dev/null | python scripts/print_bench.py "batch $bs ST=0"
  UF_ATTN_ST=1 $B --batch $bs 2>/dev/null | python scripts/print_bench.py "batch $bs ST=1"
done | grep -v train | tee -a $O/r06_run3_ab.txt
for st in 0 1 0 1; do UF_ATTN_ST=$st python scripts/history/r05_ablate.py "ST=$st"; done 2>&1 | grep -v Warn | tee $O/r06_run3_stages.txt
