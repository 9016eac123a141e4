#!/bin/bash
# round 6, run 30: Downsample second form streaming the fragment-major weight (uf_downsample_fm_fwd): parity, the four levels, the bench line and the training step against the first form
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_abi_symbols.py -m gpu -q -k "downsample or sampler or abi" 2>&1 | tail -3) | tee $O/r06_run30_pytest.txt
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_bwd.py -m gpu -q -x -k "model or uformer_B or uformer_T or traj" 2>&1 | tail -3) | tee -a $O/r06_run30_pytest.txt
for i in 1 2; do echo "=== second form, fragment-major weights"; python scripts/ubench_down.py 2>/dev/null; echo "=== second form, row-major weights"; DOWN_NO_FM=1 python scripts/ubench_down.py 2>/dev/null; echo "=== first form"; UF_VARIANT="down=1" python scripts/ubench_down.py 2>/dev/null; done | tee $O/r06_run30_down.txt
echo "=== batch 32"; python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run30_down.txt; UF_VARIANT="down=1" python scripts/ubench_down.py --batch 32 2>/dev/null | tee -a $O/r06_run30_down.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "second form (fm)"
  UF_VARIANT="down=1" python bench.py --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 5 2>/dev/null | python scripts/print_bench.py "first form      "
done | tee $O/r06_run30_ab.txt
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')" "$1"; }
for i in 1 2; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "train, second form (fm) #$i"
  UF_VARIANT="down=1" python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | show "train, first form       #$i"
done | tee -a $O/r06_run30_ab.txt
