cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do for v in 2 1; do UF_BWD_STREAMS=$v python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UF_BWD_STREAMS=$v run $r', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms', round(d['peak_mem_gb'],1), 'GB')"; done; done | tee gpurun_out/r03_train_side.txt
