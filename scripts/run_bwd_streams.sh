# A/B of the block backward with and without the side stream for the weight-gradient jobs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_bwd.py -q -x 2>&1 | tail -3 | tee $O/t_gpu.txt
for r in 1 2; do for n in 1 2; do echo "UF_BWD_STREAMS=$n"; UF_BWD_STREAMS=$n python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-250; done; done | tee $O/tb_streams.txt
