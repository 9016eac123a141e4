cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x -k "conv3x3" 2>&1 | tail -2
for i in 1 2; do python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('training', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; done | tee gpurun_out/r03_final_check.txt
