#!/bin/bash
# round 6, run 37: Upsample in its A-stationary form (uf_upsample_fm_fwd) against the tiled GEMM: output hashes, the four levels, the bench line
O=gpurun_out; mkdir -p $O
for i in 1 2; do echo "=== A-stationary"; python scripts/ubench_up.py 2>/dev/null; echo "=== tiled"; UP_NO_FM=1 python scripts/ubench_up.py 2>/dev/null; done | tee $O/r06_run37_up.txt
echo "=== batch 8 / 32"; for b in 8 32; do python scripts/ubench_up.py --batch $b 2>/dev/null | tail -1; UP_NO_FM=1 python scripts/ubench_up.py --batch $b 2>/dev/null | tail -1; done | tee -a $O/r06_run37_up.txt
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_abi_symbols.py -m gpu -q -x -k "sampler or upsample or model or abi" 2>&1 | tail -3) | tee $O/r06_run37_pytest.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "A-stationary Upsample"
  UF_VARIANT="up=1" python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "tiled Upsample       "
done | tee $O/r06_run37_ab.txt
