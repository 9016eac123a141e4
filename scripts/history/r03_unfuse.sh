cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2; do for n in 0 40 70 140 300; do echo "UF_UNFUSE_BELOW=$n run $r: $(UF_UNFUSE_BELOW=$n b)"; done; done | tee gpurun_out/r03_unfuse.txt
for n in 0 70 300; do echo "UF_STREAMS=1 UF_UNFUSE_BELOW=$n: $(UF_STREAMS=1 UF_UNFUSE_BELOW=$n b)"; done | tee -a gpurun_out/r03_unfuse.txt
