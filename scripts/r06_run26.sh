#!/bin/bash
# round 6, run 26: Downsample with three (128-wide tiles: two) K tiles of its im2col loader in flight against ab/base (the next tile only): the four levels, parity, the bench line
O=gpurun_out; mkdir -p $O
for i in 1 2; do echo "=== new"; python scripts/ubench_down.py 2>/dev/null; echo "=== base"; UFORMER_HIP_LIB=$PWD/ab/base/libuformer_hip.so python scripts/ubench_down.py 2>/dev/null; done | tee $O/r06_run26_down.txt
echo "=== batch 32"; python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run26_down.txt; UFORMER_HIP_LIB=$PWD/ab/base/libuformer_hip.so python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run26_down.txt
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "sampler or downsample or model or golden" 2>&1 | tail -3) | tee $O/r06_run26_pytest.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "new "
  UFORMER_HIP_LIB=$PWD/ab/base/libuformer_hip.so python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "base"
done | tee $O/r06_run26_ab.txt
