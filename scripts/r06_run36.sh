#!/bin/bash
# round 6, run 36: the driver's command as it is (python bench.py, no flags) on the final tree: wall clock and the line
O=gpurun_out; mkdir -p $O
s=$(date +%s); python bench.py > $O/r06_run36_bench.json 2> $O/r06_run36_bench.err; rc=$?; e=$(date +%s)
echo "rc=$rc wall=$((e-s)) s" | tee $O/r06_run36_wall.txt
python scripts/print_bench.py $O/r06_run36_bench.json | tail -4
tail -c 1500 $O/r06_run36_bench.json
