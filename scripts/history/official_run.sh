#!/bin/bash
# One command that regenerates everything under profiles/ on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash scripts/official_run.sh'
# Outputs land in gpurun_out/ (merged back by gpurun); copy the r02_* files into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee $O/r02_pytest_gpu.txt
(cd $R && python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|ok" | tail -4) | tee $O/r02_smoke.txt
cp $O/parity_model.json $O/r02_parity_model.json 2>/dev/null; cp $O/parity_ops.json $O/r02_parity_ops.json 2>/dev/null
for f in parity_grad_B_f32 parity_grad_B_bf16 parity_tiled; do cp $O/$f.json $O/r02_$f.json 2>/dev/null; done
# PMC passes first, so the bench line below can quote the measured HBM traffic of its dominant kernel (stamped with the SHA of the
# kernel sources: bench.py drops the number if the stamp does not match the sources it runs).  They and the kernel trace run with
# UF_STREAMS=1 (whole-batch launches on one stream), the configuration the library's own HIP-event timing (roofline.achieved)
# uses, so per-launch figures of the three tools describe the same launches; the headline bench at the end runs the default
# (two half-batch streams).
export UF_STREAMS=1
bash $R/scripts/pmc_passes.sh > $O/pmc_passes.log 2>&1; grep -E "^pmc. rc" $O/pmc_passes.log
(cd $R && python scripts/pmc_traffic.py $O $O/r02_pmc_traffic.json | head -8 && cp $O/r02_pmc_traffic.json $R/profiles/r02_pmc_traffic.json)
for p in A B C D E; do mv $O/pmc${p}_pmc.csv $O/r02_final_pmc${p}.csv 2>/dev/null; rm -f $O/pmc${p}_kernel_stats.csv $O/pmc${p}_gaps.txt $O/pmc${p}.log; done
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode > $O/kt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kt/kt_results.db $O/r02_final | tail -2
unset UF_STREAMS
(cd $R && for n in 1 2 3; do UF_STREAMS=$n python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UF_STREAMS=$n', round(d['value'],1), 'img/s')"; done) | tee $O/r02_streams.txt
# same-box A/B against the round-1 library (ab/r01, built from the round-1 sources) when it travelled with the snapshot
(cd $R && [ -f ab/r01/libuformer_hip.so ] && for r in 1 2; do for v in r01 r02; do if [ $v = r01 ]; then export UFORMER_HIP_LIB=$R/ab/r01/libuformer_hip.so UF_ALLOW_OLDER_LIB=1; else unset UFORMER_HIP_LIB UF_ALLOW_OLDER_LIB; fi; python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v run $r', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; done; done; unset UFORMER_HIP_LIB UF_ALLOW_OLDER_LIB) | tee $O/r02_ab_vs_r01.txt
(cd $R && python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1) | tee $O/r02_train_step.json
(cd $R && python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 --sink 2>/dev/null | tail -1) | tee $O/r02_train_step_sink.json
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r02_train_final | tail -2
(cd $R && python scripts/ubench_train.py all 2>/dev/null | grep -v amdgpu.ids) > $O/r02_ubench_train.txt; tail -1 $O/r02_ubench_train.txt
(cd $R && python bench.py --error-budget --kernels-json $O/r02_kernels_hip_events.json > $O/r02_bench.json 2> $O/bench.err; cut -c1-700 $O/r02_bench.json)
(cd $R && python bench.py --img 1280 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode 2>/dev/null | cut -c1-400) | tee $O/r02_bench_720p.json
ls -la $O
