#!/bin/bash
# round 6, run 23: PipelinedForward test output in full; bench line with modes.pipelined (lanes as shipped: one pool)
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "pipelined" 2>&1 | tail -60) > $O/r06_run23_pytest.txt; tail -5 $O/r06_run23_pytest.txt
for i in 1 2; do python scripts/pipelined_bench.py 2>/dev/null | tail -1; done | tee $O/r06_run23_pipelined.txt
python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p 2>$O/r06_run23_bench.err | tail -1 > $O/r06_run23_bench.json; tail -c 700 $O/r06_run23_bench.json
