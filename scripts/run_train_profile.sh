R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/t_gpu.txt; cat $O/t_gpu.txt
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | tee $O/tb.txt
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 --sink 2>&1 | tail -1 | tee -a $O/tb.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/kt_train.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kt/kt_results.db $O/r02_train_v5 | tail -2
