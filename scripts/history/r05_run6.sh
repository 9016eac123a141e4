#!/bin/bash
# round 5, run 6: evidence for the two-window question -- attn_block as shipped vs the same kernel with NO weight-fragment loads (UF_ABL=6: the limit
# any weight-sharing tile can approach): phase stamps and SQ counters of both, on one LeWin block per deep-stage shape; then the stream-count sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L6=$R/ab/abl6/libuformer_hip.so
{
echo "== per-kernel time, one LeWin block per shape (HIP events, 30 launches) =="
python $R/scripts/r05_ablate.py "shipped"
UFORMER_HIP_LIB=$L6 UF_ALLOW_OLDER_LIB=1 python $R/scripts/r05_ablate.py "no weight loads"
echo "== phase stamps (shader cycles, wave 0 of sampled windows): shipped =="
python $R/scripts/ubench.py stamps | grep -E "^C=|block +(64|192|128) *:" | grep -v "start +[0-9]{1,6} \|"
echo "== phase stamps: no weight loads =="
UFORMER_HIP_LIB=$L6 UF_ALLOW_OLDER_LIB=1 python $R/scripts/ubench.py stamps | grep -E "^C=|block +(64|192|128) *:"
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_attn2w.txt
run() { name=$1; lib=$2; shift 2; UFORMER_HIP_LIB=$lib UF_ALLOW_OLDER_LIB=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/$name -o $name -- python $R/scripts/r05_ablate.py $name > $O/$name.log 2>&1; echo $name rc=$?; python $R/scripts/rocprof_summary.py /tmp/$name/${name}_results.db $O/$name | tail -1; }
D=$R/uformer_amd/lib/libuformer_hip.so
run r05_pmcA_shipped $D SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run r05_pmcB_shipped $D SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run r05_pmcA_noweights $L6 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run r05_pmcB_noweights $L6 SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
cd $R
for s in 1 2 3 4; do UF_STREAMS=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-train-mode --no-720p 2>/dev/null | python scripts/print_bench.py "UF_STREAMS=$s"; done | tee $O/r05_run6_streams.txt
ls $O | grep r05_pmc
