# Round 3: XCD-aware workgroup order in linear_wgrad2: tests, microbench, training step A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x -k "wgrad or block or model" 2>&1 | tail -2
{
echo "== XCD-aware (default)"; python scripts/ubench_train.py wgrad 2>/dev/null | tail -1
echo "== UF_WGRAD_XCD=0"; UF_WGRAD_XCD=0 python scripts/ubench_train.py wgrad 2>/dev/null | tail -1
} | tee $O/r03_wgrad_xcd.txt
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{ tb "XCD-aware wgrad"; UF_WGRAD_XCD=0 tb "UF_WGRAD_XCD=0"; tb "XCD-aware again"; } | tee -a $O/r03_wgrad_xcd.txt
