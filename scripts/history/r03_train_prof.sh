# Round 3: vectorised im2col / col2im check + per-kernel profile of the training step in its stored-intermediates form
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $r', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms', round(d['peak_mem_gb'],1), 'GB')"; done | tee $O/r03_train_vec.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r03_train_stored | tail -2
