#!/bin/bash
# round 6, run 9: functional run of the N-rank bench path on the 1-GPU box (2 ranks share the GPU, gloo) with the ring and with the direct exchange; not a measurement
O=gpurun_out; mkdir -p $O
export UF_BENCH_BACKEND=gloo UF_BENCH_SHARE_GPU=1
A="--steps 3 --warmup 1 --repeats 2 --no-cpu-baseline --no-vendor-baseline --no-other-modes --no-720p --train-batch 8 --train-steps 2"
timeout 600 python bench.py --gpus 2 $A > $O/r06_two_ranks_ring.json 2> $O/r06_two_ranks_ring.err; echo "ring rc=$?"; tail -c 600 $O/r06_two_ranks_ring.err
timeout 600 python bench.py --gpus 2 $A --exchange direct --exchange-payload bf16 > $O/r06_two_ranks_direct.json 2> $O/r06_two_ranks_direct.err; echo "direct rc=$?"; tail -c 1500 $O/r06_two_ranks_direct.err
python - <<'P'
import json
for t in ("ring", "direct"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06_two_ranks_{t}.json").read().splitlines() if l.startswith("{")][-1])
        tr = d["modes"]["train"]
        print(t, "ranks", len(d["config"]["rank_devices"]), "train", {k: tr.get(k) for k in ("images_per_s", "gradient_exchange", "exchange_buckets", "exchange_bytes_per_step", "exchange_wire_bytes_per_rank_per_step", "loss_is_finite", "error")})
    except Exception as e:
        print(t, "no line:", e)
P
