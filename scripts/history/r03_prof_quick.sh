cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r03_train_stored5 | tail -1
grep -E "conv3x3|_proj_kernel" $O/r03_train_stored5_kernel_stats.csv | sed 's/uf::(anonymous namespace):://' | awk -F'"' '{print substr($2,1,60), $3}'
