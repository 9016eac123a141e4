#!/bin/bash
# round 4, GPU call 21: every kernel file built with -fno-slp-vectorize (ab/noslp_all) against the shipped build (only attn_block without SLP)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bi() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode --no-720p 2>/dev/null | python scripts/print_bench.py "$1"; }
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms')"; }
{
for r in 1 2 3; do bi "shipped build run $r"; UFORMER_HIP_LIB=$R/ab/noslp_all/libuformer_hip.so bi "no SLP anywhere run $r"; done
for r in 1 2; do echo "train shipped build run $r: $(tb)"; echo "train no SLP anywhere run $r: $(UFORMER_HIP_LIB=$R/ab/noslp_all/libuformer_hip.so tb)"; done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run21.txt
