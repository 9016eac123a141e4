# Round 3: native pack in the stored form + fused depthwise backward: full backward tests, A/B, per-kernel profile
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_bwd.py -m gpu -q -x 2>&1 | tail -3
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms', round(d['peak_mem_gb'],1), 'GB')"; }
{
tb "default (native pack)"
UF_PY_PACK=1 tb "UF_PY_PACK=1"
UF_DWBWD_ROWS=2 tb "UF_DWBWD_ROWS=2"
tb "default again"
} | tee $O/r03_train_pack.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktt -o ktt -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/ktt.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/ktt/ktt_results.db $O/r03_train_stored2 | tail -2
