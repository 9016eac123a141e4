R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/kt_train.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kt/kt_results.db $O/r02_train_v6 | tail -2
