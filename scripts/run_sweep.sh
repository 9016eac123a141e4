# batch-size sweep of the inference bench and the training step on one GPU (profiles/${ROUND_TAG:-r04}_batch_sweep.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for b in 4 8 16 32 64; do python bench.py --batch $b --no-cpu-baseline --no-other-modes --no-train-mode --no-720p 2>/dev/null | python scripts/print_bench.py "inference batch $b"; done | tee $O/${ROUND_TAG:-r04}_batch_sweep.txt
for b in 8 16 32 64; do python scripts/train_bench.py --batch $b --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('training batch $b', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms', round(d['peak_mem_gb'],1), 'GB')"; done | tee -a $O/${ROUND_TAG:-r04}_batch_sweep.txt
