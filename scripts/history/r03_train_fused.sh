# Round 3: fused depthwise backward (uf_dwconv3x3_bwd), single-buffer GEMM launches for K <= 64, 3 backward streams: tests + A/B
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x -k "dwconv or block or model or linear" 2>&1 | tail -3
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms', round(d['peak_mem_gb'],1), 'GB')"; }
{
tb "default (fused dw bwd, LDS1)"
UF_DW_BWD_FUSED=0 tb "UF_DW_BWD_FUSED=0"
UF_GEMM_LDS2=1 tb "UF_GEMM_LDS2=1"
UF_BWD_STREAMS=3 tb "UF_BWD_STREAMS=3"
UF_DWBWD_VEC=8 tb "UF_DWBWD_VEC=8"
UF_DWBWD_BLOCKS=512 tb "UF_DWBWD_BLOCKS=512"
tb "default again"
} | tee $O/r03_train_fused.txt
