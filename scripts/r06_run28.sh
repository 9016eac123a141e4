#!/bin/bash
# round 6, run 28: Downsample second form with the K loop fully unrolled (a real 7-deep weight ring) and eight staging chunks in flight
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "downsample or sampler" 2>&1 | tail -3) | tee $O/r06_run28_pytest.txt
for i in 1 2; do echo "=== second form"; python scripts/ubench_down.py 2>/dev/null; echo "=== first form"; UF_VARIANT="down=1" python scripts/ubench_down.py 2>/dev/null; done | tee $O/r06_run28_down.txt
echo "=== batch 32"; python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run28_down.txt; UF_VARIANT="down=1" python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run28_down.txt
echo "=== batch 8 (one half-batch part)"; python scripts/ubench_down.py --batch 8 2>/dev/null | tee -a $O/r06_run28_down.txt; UF_VARIANT="down=1" python scripts/ubench_down.py --batch 8 2>/dev/null | tee -a $O/r06_run28_down.txt
