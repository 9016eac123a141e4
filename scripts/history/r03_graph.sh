cd $GRAFT_REPO_ROOT
export UF_TRAIN_RECOMPUTE=0
{ timeout 300 python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 --graph 2>&1 | tail -4 | cut -c1-600; python scripts/train_bench.py --batch 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-300; } | tee gpurun_out/r03_graph.txt
