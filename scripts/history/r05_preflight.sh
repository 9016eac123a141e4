#!/bin/bash
# pre-flight of the final tree: the whole GPU suite, then the driver's default bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -12) | tee $O/r05_preflight_pytest.txt
(time python bench.py > $O/r05_preflight_bench.json 2> $O/r05_preflight_bench.err) 2>&1 | tail -3
tail -c 2500 $O/r05_preflight_bench.json; tail -5 $O/r05_preflight_bench.err
