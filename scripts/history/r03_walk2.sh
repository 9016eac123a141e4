# Round 3: the fused depthwise backward walking along x: tests + microbench + training step
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x -k "dwconv or block or model" 2>&1 | tail -3
{
echo "== walking stencils (default)"; python scripts/ubench_train.py stencil 2>/dev/null | grep -E "dwconv_bwd_fused|^\{"
} | tee $O/r03_walk2.txt
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{ tb "walking fwd + bwd"; UF_DWCONV_WALK=0 tb "UF_DWCONV_WALK=0"; tb "walking again"; } | tee -a $O/r03_walk2.txt
