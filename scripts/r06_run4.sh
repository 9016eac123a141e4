#!/bin/bash
# round 6, run 4: single-operand-tile attn_block (UF_ATTN_ST=1) with the whole batch on ONE stream, and the SQ counters of both forms
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-vendor-baseline --no-train-mode --no-720p --no-other-modes"
cd $R
for i in 1 2; do
  UF_STREAMS=1 UF_ATTN_ST=0 $B 2>/dev/null | python scripts/print_bench.py "one stream, two tiles (ST=0) #$i"
  UF_STREAMS=1 UF_ATTN_ST=1 $B 2>/dev/null | python scripts/print_bench.py "one stream, single tile (ST=1) #$i"
done | grep -v train | tee $O/r06_run4_ab.txt
cd /tmp; export TMPDIR=/tmp UF_STREAMS=1
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/$name -o $name -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-modes --no-train-mode --no-720p --no-vendor-baseline --repeats 1 > $O/$name.log 2>&1; echo $name rc=$?; python $R/scripts/rocprof_summary.py /tmp/$name/${name}_results.db $O/$name | tail -1; }
for st in 0 1; do
  export UF_ATTN_ST=$st
  run r06_pmcA_st$st SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run r06_pmcB_st$st SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
done
grep -h "attn_block_kernel<uf::bf16, 256, 256" $O/r06_pmc?_st?_pmc.csv | cut -c1-400
rm -f $O/r06_pmc*.log
