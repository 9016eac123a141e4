# Round 3: is the depthwise stencil bound by the 3x re-read of its input through L1/L2?  Timing ablation: one column tap instead of three
cd $GRAFT_REPO_ROOT
{
echo "== default"; python scripts/ubench_train.py stencil 2>/dev/null | grep -E "dwconv_pre_gelu|dwconv_bwd_fused|dwconv_plain|^\{"
echo "== one column tap (ab/kx1: wrong results, timing only; forward kernels only)"; UFORMER_HIP_LIB=ab/kx1/libuformer_hip.so python scripts/ubench_train.py stencil 2>/dev/null | grep -E "dwconv_pre_gelu|dwconv_plain|^\{"
echo "== training batch 64 with the stored form (cap lifted)"; python scripts/train_bench.py --batch 64 --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-330
} | tee gpurun_out/r03_kx_abl.txt
