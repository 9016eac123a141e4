# same-box A/B of an environment switch: bash scripts/run_ab_env.sh VAR=value   (bench with and without it, 3 runs each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; KV=$1
env $KV python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -x 2>&1 | tail -2 | tee $O/ab_env_tests.txt
for r in 1 2 3; do
python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python scripts/print_bench.py "base run $r"
env $KV python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python scripts/print_bench.py "$KV run $r"
done | tee $O/ab_env.txt
env $KV UF_STREAMS=1 python bench.py --no-cpu-baseline --no-f32-mode --kernels-json $O/k_var.json 2>/dev/null | python scripts/print_bench.py "$KV one stream"
