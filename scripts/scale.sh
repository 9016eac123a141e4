#!/bin/bash
# Scaling runs on an N-GPU node: `python bench.py --gpus N` launches itself as one process per GPU (torch.distributed.run, RCCL over xGMI).
# Every line carries the inference metric (batch-sharded replicas, no collective), modes.train (the bucketed gradient all-reduce overlapped
# with the reverse sweep: exchange_buckets / exchange_exposed_ms_per_step) and config.rank_devices (all-gathered over RCCL: proof N ranks met).
#   bash scripts/scale.sh [outdir] [gpus...]        default: gpurun_out/scale  1 2 4 8
R=$(cd "$(dirname "$0")/.." && pwd); O=${1:-$R/gpurun_out/scale}; shift; NS=${@:-1 2 4 8}; mkdir -p $O; cd $R
for n in $NS; do
  python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n$n.json 2> $O/bench_n$n.err
  python scripts/print_bench.py $O/bench_n$n.json 2>/dev/null | tail -3
done
