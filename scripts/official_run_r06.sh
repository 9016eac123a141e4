#!/bin/bash
# Everything under profiles/r06_final_* / r06_bench* / r06_train_* / r06_parity_* / r06_pmc_* from ONE command on an MI355X box:
#   gpurun --timeout 2700 -- 'bash scripts/official_run_r06.sh'
# Outputs land in gpurun_out/ (merged back by gpurun); copy the r06_* files into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp ROUND_TAG=r06
(cd $R && python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee $O/r06_pytest_gpu.txt
(cd $R && python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|ok" | tail -5) | tee $O/r06_smoke.txt
for f in parity_model parity_ops parity_traj_f32 parity_traj_f16 parity_traj_bf16 parity_grad_B_f32 parity_grad_B_bf16 parity_grad_B_f16 parity_grad_T_f32 parity_grad_T_bf16 parity_grad_T_f16 parity_tiled; do cp $O/$f.json $O/r06_$f.json 2>/dev/null; done
# PMC passes first (the bench line quotes the measured HBM traffic of its dominant kernel, stamped with the SHA of the kernel sources).  They and the
# kernel trace run with UF_STREAMS=1 (whole-batch launches on one stream), the configuration the library's own HIP-event timing uses.
export UF_STREAMS=1
bash $R/scripts/pmc_passes.sh > $O/pmc_passes.log 2>&1; grep -E "^pmc. rc" $O/pmc_passes.log
(cd $R && python scripts/pmc_traffic.py $O $O/r06_pmc_traffic.json | head -8 && cp $O/r06_pmc_traffic.json $R/profiles/r06_pmc_traffic.json)   # pmc_traffic.py also reads the SQ passes A / B (MFMA pipe busy, VALU active per symbol)
for p in A B C D E; do mv $O/pmc${p}_pmc.csv $O/r06_final_pmc${p}.csv 2>/dev/null; rm -f $O/pmc${p}_kernel_stats.csv $O/pmc${p}_gaps.txt $O/pmc${p}.log; done
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p > $O/kt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kt/kt_results.db $O/r06_final | tail -2
unset UF_STREAMS
bash $R/scripts/pmc_train.sh > $O/pmc_train.log 2>&1; grep -E "^pmcT. rc|all kernels" $O/pmc_train.log; cp $O/r06_pmc_traffic_train.json $R/profiles/r06_pmc_traffic_train.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 3 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r06_train_final | tail -2
UF_BWD_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/kts -o kts -- python $R/scripts/train_bench.py --batch 32 --steps 3 --warmup 1 > $O/kts.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kts/kts_results.db $O/r06_train_serial | tail -2
(cd $R && python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --dtype bf16 --kernels-json $O/r06_train_kernels_hip_events.json 2>/dev/null | tail -1;
 cd $R && python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --dtype bf16 --no-fused-attn 2>/dev/null | tail -1;
 cd $R && python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --dtype f16 2>/dev/null | tail -1) | tee $O/r06_train_step.json
(cd $R && bash scripts/run_sweep.sh 2>/dev/null | tail -12)
(cd $R && python bench.py --error-budget --kernels-json $O/r06_kernels_hip_events.json > $O/r06_bench.json 2> $O/bench.err; tail -c 1800 $O/r06_bench.json)
ls -la $O | tail -40
