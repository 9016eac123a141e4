#!/bin/bash
# round 6, run 1: the fused attention-half training forward (uf_lewin_attn_train_fwd): parity subset, A/B of the training step, packed-f16 VALU rates
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -x -q 2>&1 | tail -8) | tee $O/r06_run1_pytest.txt
for i in 1 2; do
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --no-fused-attn 2>/dev/null | python scripts/print_bench.py "op-by-op attn fwd #$i" | tee -a $O/r06_run1_ab.txt
  python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | python scripts/print_bench.py "fused attn fwd    #$i" | tee -a $O/r06_run1_ab.txt
done
python scripts/train_bench.py --batch 32 --steps 2 --warmup 1 --kernels-json $O/r06_run1_train_kernels.json > /dev/null 2>&1
(cd scripts/ubench_hip && timeout 300 ./valu_rate) > $O/r06_run1_valu_rate.txt 2>&1
tail -22 $O/r06_run1_valu_rate.txt
