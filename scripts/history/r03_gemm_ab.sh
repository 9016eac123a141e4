# Round 3: deeper load-ahead in gemm_kernel (2 K tiles) and linear_wgrad2 (2-3 token steps): op tests, microbench A/B, training step A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bwd.py -m gpu -q -x -k "linear or gemm or sampler or qkv or wgrad or projection or block or model" 2>&1 | tail -3
{
echo "== gemm, load-ahead 2 K tiles (default)"; python scripts/ubench_train.py gemm 2>/dev/null
echo "== gemm, load-ahead 1 K tile (ab/pd1)"; UFORMER_HIP_LIB=ab/pd1/libuformer_hip.so python scripts/ubench_train.py gemm 2>/dev/null
for d in 2 1 3; do echo "== wgrad, UF_WGRAD_DEPTH=$d"; UF_WGRAD_DEPTH=$d python scripts/ubench_train.py wgrad 2>/dev/null; done
} > $O/r03_gemm_ab.txt
grep "^{\|^==" $O/r03_gemm_ab.txt
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{
tb "default (gemm 2 tiles ahead, wgrad depth 2)"
UF_WGRAD_DEPTH=1 tb "UF_WGRAD_DEPTH=1"
UF_WGRAD_DEPTH=3 tb "UF_WGRAD_DEPTH=3"
UFORMER_HIP_LIB=ab/pd1/libuformer_hip.so tb "gemm 1 tile ahead (ab/pd1)"
tb "default again"
} | tee $O/r03_train_depth.txt
