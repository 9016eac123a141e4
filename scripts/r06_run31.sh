#!/bin/bash
# round 6, run 31: the head on the matrix pipe with split operands (uf_output_proj_t_fwd) against the f32 form: parity, time, the bench line (UF_VARIANT="head=2" = the f32 form)
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_abi_symbols.py -m gpu -q -k "output_proj or sampler or abi or stem" 2>&1 | tail -5) | tee $O/r06_run31_pytest.txt
(python scripts/ubench_head.py; python scripts/ubench_head.py --batch 8; python scripts/ubench_head.py --batch 32) 2>/dev/null | tee $O/r06_run31_head.txt
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_bwd.py tests/test_gpu_traj.py -m gpu -q -x -k "model or uformer_B or uformer_T or traj" 2>&1 | tail -3) | tee -a $O/r06_run31_pytest.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-pipelined --repeats 5 2>/dev/null | python scripts/print_bench.py "split-operand head"
  UF_VARIANT="head=2" python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-pipelined --repeats 5 2>/dev/null | python scripts/print_bench.py "f32 head          "
done | tee $O/r06_run31_ab.txt
python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-pipelined --repeats 3 2>/dev/null | tail -c 1200 | tee -a $O/r06_run31_ab.txt
