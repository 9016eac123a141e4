#!/bin/bash
# round 4, GPU call 13: the training step with every kernel on ONE stream (UF_BWD_STREAMS=1 UF_STREAMS=1): per-kernel durations without the
# overlap that inflates them in the two-stream step (linear_wgrad 245 us in the step vs 105 us alone, profiles/r04_run11.txt / r04_run12.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
tb() { python $R/scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms', 'host', round(d.get('host_enqueue_ms_per_step',0),1))"; }
{
echo "train, default streams: $(tb)"
echo "train, one stream: $(UF_BWD_STREAMS=1 UF_STREAMS=1 tb)"
echo "train, side stream only for weight gradients (UF_STREAMS=1): $(UF_STREAMS=1 tb)"
UF_BWD_STREAMS=1 UF_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/kts -o kts -- python $R/scripts/train_bench.py --batch 32 --steps 2 --warmup 1 > $O/kts.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/kts/kts_results.db $O/r04_train_serial | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run13.txt
