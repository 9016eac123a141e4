R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bwd.py -q -x 2>&1 | tail -3 > $O/t_gpu.txt; cat $O/t_gpu.txt
UF_STREAMS=1 python bench.py --no-cpu-baseline --no-f32-mode --kernels-json $O/k_new.json 2>/dev/null | python scripts/print_bench.py new
