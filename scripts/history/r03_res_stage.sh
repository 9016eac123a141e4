# Round 3: LDS-staged f32 residual stores of the GEMM (E_RES / E_RES_WINREV): tests + training step
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bwd.py tests/test_gpu_model.py -m gpu -q -x -k "linear or gemm or residual or block or model or golden" 2>&1 | tail -3
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{ tb "staged residual stores"; tb "again" "--kernels-json $O/r03_train_kernels2.json"; } | tee $O/r03_train_res_stage.txt
python scripts/ubench_train.py gemm 2>/dev/null | grep "fc2_residual" | tee $O/r03_res_stage_ubench.txt
