#!/bin/bash
# round 6, run 24: hardware queues.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams beyond that share one and block each other's heads.
# The inference loop (2 streams per forward) and PipelinedForward (1-3 forwards in flight) at 2 / 4 (default) / 8 queues.
O=gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "pipelined" 2>&1 | tail -3) | tee $O/r06_run24_pytest.txt
for q in 4 8 2 4 8; do echo "=== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python scripts/pipelined_bench.py 2>/dev/null | tail -1; done | tee $O/r06_run24_queues.txt
for q in 4 8; do echo "=== GPU_MAX_HW_QUEUES=$q UF_STREAMS=3"; GPU_MAX_HW_QUEUES=$q UF_STREAMS=3 python scripts/pipelined_bench.py 2>/dev/null | tail -1; done | tee -a $O/r06_run24_queues.txt
for q in 4 8; do echo "=== GPU_MAX_HW_QUEUES=$q train"; GPU_MAX_HW_QUEUES=$q python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-200; done | tee -a $O/r06_run24_queues.txt
