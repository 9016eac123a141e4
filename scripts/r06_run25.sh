#!/bin/bash
# round 6, run 25: unequal half-batch parts (the two parts of a forward run in lockstep and sit in the 64-window bottleneck together): images of part 0 = 8 (shipped) / 9 / 10 / 11 / 12 / 6 of 16
O=gpurun_out; mkdir -p $O
for r in 1 2; do for v in 8 9 10 11 12 6; do echo -n "split0=$v  "; UF_VARIANT="split0=$v" python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "bf16 batch 16"; done; done | tee $O/r06_run25_split.txt
