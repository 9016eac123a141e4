#!/bin/bash
# round 6, run 27: Downsample second form (input patch staged in LDS once, fragments of every tap read from it, weights L2 -> registers) against the im2col-loader GEMM (UF_VARIANT="down=1")
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "downsample or sampler" 2>&1 | tail -15) | tee $O/r06_run27_pytest.txt
for i in 1 2; do echo "=== second form"; python scripts/ubench_down.py 2>/dev/null; echo "=== first form"; UF_VARIANT="down=1" python scripts/ubench_down.py 2>/dev/null; done | tee $O/r06_run27_down.txt
echo "=== batch 32"; python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run27_down.txt; UF_VARIANT="down=1" python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run27_down.txt
echo "=== batch 1 f16"; python scripts/ubench_down.py --batch 1 --dtype f16 2>/dev/null | tee -a $O/r06_run27_down.txt; UF_VARIANT="down=1" python scripts/ubench_down.py --batch 1 --dtype f16 2>/dev/null | tee -a $O/r06_run27_down.txt
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3) | tee -a $O/r06_run27_pytest.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "second form"
  UF_VARIANT="down=1" python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "first form "
done | tee $O/r06_run27_ab.txt
