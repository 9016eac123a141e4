# PMC passes of the bench workload, counters only (+ --kernel-trace), one pass per counter group (SQ: 8 slots; FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).  Run through scripts/official_run.sh.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/$name -o $name -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-train-mode --no-720p --repeats 1 > $R/gpurun_out/$name.log 2>&1; echo $name rc=$?; python $R/scripts/rocprof_summary.py /tmp/$name/${name}_results.db $R/gpurun_out/$name | tail -1; }
run pmcA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmcB SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
run pmcC FETCH_SIZE
run pmcD WRITE_SIZE
run pmcE TCC_HIT_sum TCC_MISS_sum
ls -la $R/gpurun_out
