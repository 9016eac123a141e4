#!/bin/bash
# round 6, run 21: uformer_amd.infer.PipelinedForward (1 / 2 / 3 forwards of successive batches in flight, each on its own stream with its own lane of side streams) against the eager loop,
# with the library's default two half-batch parts per forward and with the whole batch on one stream per forward (UF_STREAMS=1)
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "pipelined or graph or stream" 2>&1 | tail -3) | tee $O/r06_run21_pytest.txt
for i in 1 2; do
  python scripts/pipelined_bench.py 2>/dev/null | tail -1
  UF_STREAMS=1 python scripts/pipelined_bench.py 2>/dev/null | tail -1
  UF_STREAMS=3 python scripts/pipelined_bench.py 2>/dev/null | tail -1
done | tee $O/r06_run21_pipelined.txt
python scripts/pipelined_bench.py --batch 4 --steps 80 2>/dev/null | tail -1 | tee -a $O/r06_run21_pipelined.txt
python scripts/pipelined_bench.py --batch 1 --steps 200 2>/dev/null | tail -1 | tee -a $O/r06_run21_pipelined.txt
python scripts/pipelined_bench.py --dtype f16 2>/dev/null | tail -1 | tee -a $O/r06_run21_pipelined.txt
