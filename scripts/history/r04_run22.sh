#!/bin/bash
# round 4, GPU call 22: the default driver command after the last bench.py / train.py edits (smoke + full bench line)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ python __graft_entry__.py --smoke 2>&1 | grep smoke | tail -4; python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc=$?"; tail -c 1800 $O/r04_bench_final.json; } 2>&1 | grep -v amdgpu.ids | tee $O/r04_run22.txt
