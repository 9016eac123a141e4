#!/bin/bash
# Scaling runs for the day an 8-GPU node is available (VERDICT r02 "next" 8).  One process per GPU under torch.distributed.run,
# RCCL over xGMI; the inference bench shards the batch (no data-path collective, weak scaling), the training step all-reduces its
# gradient buckets overlapped with the reverse sweep.  Every JSON line carries config.rank_devices = the per-rank device list
# all-gathered over RCCL, so the record itself proves N ranks met.
#   bash scripts/scale.sh [outdir] [gpus...]        default: gpurun_out/scale  1 2 4 8
R=$(cd "$(dirname "$0")/.." && pwd); O=${1:-$R/gpurun_out/scale}; shift; NS=${@:-1 2 4 8}; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29511
for n in $NS; do
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
    python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 > $O/train_n1.json 2> $O/train_n1.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline \
      > $O/bench_n$n.json 2> $O/bench_n$n.err; PORT=$((PORT + 1))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT scripts/train_bench.py --batch 32 --steps 3 --warmup 2 \
      > $O/train_n$n.json 2> $O/train_n$n.err; PORT=$((PORT + 1))
  fi
  python - "$O" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
for kind in ("bench", "train"):
    try:
        line = [l for l in open(f"{o}/{kind}_n{n}.json") if l.startswith("{")][-1]
        d = json.loads(line)
        devs = d.get("config", {}).get("rank_devices") or d.get("rank_devices")
        print(f"{kind} n={n}: {d['value']:.1f} img/s, {d['ms_per_step']:.2f} ms/step, ranks met: {len(devs) if devs else '?'}")
    except Exception as e:   # noqa: BLE001
        print(f"{kind} n={n}: no result ({e})")
PY
done
