# Round 3: 256 x 256 GEMM tiles (128 x 128 per wave, AGPR accumulators): tests, microbench A/B, training step A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm_256 or linear" 2>&1 | tail -3
{
echo "== gemm, 256x256 tiles where they apply (default)"; python scripts/ubench_train.py gemm 2>/dev/null
echo "== gemm, UF_GEMM_BIG=0"; UF_GEMM_BIG=0 python scripts/ubench_train.py gemm 2>/dev/null
echo "== gemm, UF_GEMM_BIG_K=128"; UF_GEMM_BIG_K=128 python scripts/ubench_train.py gemm 2>/dev/null
} > $O/r03_gemm_big.txt
grep "^{\|^==" $O/r03_gemm_big.txt
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{
tb "default (256x256 tiles)"
UF_GEMM_BIG=0 tb "UF_GEMM_BIG=0"
UF_GEMM_BIG_K=128 tb "UF_GEMM_BIG_K=128"
tb "default again"
} | tee $O/r03_train_big.txt
