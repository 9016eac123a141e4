R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_bwd.py -q -k "streaming_helpers or conv3x3_bwd or fused_training or merged_output" 2>&1 | grep -E "Error|error|passed|failed|assert" | head -30 > $O/t_bwd.txt; cat $O/t_bwd.txt
for v in default st_occ4; do echo "== stencil variant $v"; if [ $v != default ]; then export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi; python scripts/ubench_train.py stencil 2>&1 | grep -E "dwconv_(plain|pre|mul)|^\{"; unset UFORMER_HIP_LIB; done | tee $O/ub_stencil.txt
UF_WGRAD_TARGET=512 python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | tee $O/tb.txt
