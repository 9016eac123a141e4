cd $GRAFT_REPO_ROOT
for m in "" "UF_TRAIN_RECOMPUTE=1"; do env $m python scripts/train_bench.py --batch 32 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms  host enqueue', round(d['host_enqueue_ms_per_step'],1), 'ms')"; done | tee gpurun_out/r03_host.txt
