#!/bin/bash
# round 4, GPU call 11: wide column sum (A/B) and the per-kernel table of a training step (HIP events, GB/s against the algorithmic bytes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
tb() { python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', round(d.get('ms_per_step',0),2), 'ms')"; }
{
echo "== pytest (wgrad, layernorm, determinism)"; python -m pytest tests/test_gpu_bwd.py -m gpu -q -k "wgrad or layernorm or determin or block or model" 2>&1 | tail -3
for r in 1 2; do echo "train colsum v1 run $r: $(UF_COLSUM_V1=1 tb)"; echo "train colsum wide run $r: $(tb)"; done
python scripts/train_bench.py --batch 32 --steps 3 --warmup 2 --kernels-json $O/r04_train_kernels_run11.json > /dev/null 2>&1
python - <<'P'
import json
d=json.load(open('gpurun_out/r04_train_kernels_run11.json'))
ks=d if isinstance(d,list) else d.get('kernels',d)
tot=sum(k['ms_per_step'] for k in ks)
print('kernel-time sum per step %.2f ms'%tot)
for k in sorted(ks,key=lambda k:-k['ms_per_step'])[:70]:
    print('%-52s %8.3f ms %5d launches %8.1f us  %7.1f TF %7.0f GB/s'%(k['kernel'][:52],k['ms_per_step'],k['launches_per_step'],1e3*k['ms_per_step']/max(1,k['launches_per_step']),k.get('tflops',0),k.get('gbs',0)))
P
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_run11.txt
