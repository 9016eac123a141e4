R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_bwd.py -q -x 2>&1 | tail -4 > $O/t_gpu.txt; cat $O/t_gpu.txt
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-260 | tee $O/tb.txt
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-260 | tee -a $O/tb.txt
