# Round 3: walking output_proj + 4-wave head weight gradient: tests, inference A/B, training step
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bwd.py tests/test_gpu_model.py -m gpu -q -x -k "proj or conv3x3 or golden or model" 2>&1 | tail -3
ib() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-modes --no-train-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; }
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{
ib "inference, walking output_proj"; UF_OUTPROJ_WALK=0 ib "inference, UF_OUTPROJ_WALK=0"; ib "inference, walking again"; UF_OUTPROJ_WALK=0 ib "inference, UF_OUTPROJ_WALK=0 again"
tb "training"; tb "training again"
} | tee $O/r03_outproj.txt
