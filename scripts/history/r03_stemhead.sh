# Round 3: stem / head backward kernels rewritten (one pass over the wide tensor, sliding windows): tests + training step + profile
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x -k "conv3x3 or model" 2>&1 | tail -3
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"; }
{ tb "stem/head backward rewritten"; tb "again"; } | tee $O/r03_stemhead.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r03_train_stored4 | tail -1
grep -E "conv3x3|input_proj|output_proj" $O/r03_train_stored4_kernel_stats.csv | cut -c1-80,200-400
