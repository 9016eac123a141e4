# HBM traffic of the training step: FETCH_SIZE and WRITE_SIZE in separate counter-only passes (+ --kernel-trace) of one training step
# at batch 32; scripts/pmc_traffic_train.py turns the two CSVs into profiles/${ROUND_TAG:-r04}_pmc_traffic_train.json (read by bench.py's train mode).
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/$name -o $name -- python $R/scripts/train_bench.py --batch 32 --steps 1 --warmup 1 > $O/$name.log 2>&1; echo $name rc=$?; python $R/scripts/rocprof_summary.py /tmp/$name/${name}_results.db $O/$name | tail -1; }
run pmcTC FETCH_SIZE
run pmcTD WRITE_SIZE
cd $R && python scripts/pmc_traffic_train.py $O $O/${ROUND_TAG:-r04}_pmc_traffic_train.json | head -24
