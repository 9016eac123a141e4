#!/bin/bash
# One command that regenerates everything under profiles/ on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash scripts/official_run.sh'
# Outputs land in gpurun_out/ (merged back by gpurun); copy the r01_* files into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee $O/r01_pytest_gpu.txt
(cd $R && python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|ok" | tail -4) | tee $O/r01_smoke.txt
# PMC passes first, so the bench line below can quote the measured HBM traffic of its dominant kernel.  They and the kernel
# trace run with UF_STREAMS=1 (whole-batch launches on one stream), the configuration the library's own HIP-event timing
# (roofline.achieved) uses, so per-launch figures of the three tools describe the same launches; the headline bench at
# the end runs the default (two half-batch streams).
export UF_STREAMS=1
bash $R/scripts/pmc_passes.sh > $O/pmc_passes.log 2>&1; tail -6 $O/pmc_passes.log | head -4
python $R/scripts/pmc_traffic.py $O $O/r01_pmc_traffic.json | head -8 && cp $O/r01_pmc_traffic.json $R/profiles/r01_pmc_traffic.json
for p in A B C D; do mv $O/pmc${p}_pmc.csv $O/r01_final_pmc${p}.csv; rm -f $O/pmc${p}_kernel_stats.csv $O/pmc${p}.log; done
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kt/kt_results.db $O/r01_final | tail -2
unset UF_STREAMS
(cd $R && python bench.py --kernels-json $O/r01_kernels_hip_events.json > $O/r01_bench.json 2> $O/bench.err; cut -c1-600 $O/r01_bench.json)
ls -la $O
