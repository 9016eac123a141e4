#!/bin/bash
# same-box A/B: attn_block register bounds (ab/lb4: C=32 at 5 waves/SIMD, C=64 at 4; ab/lb4b: C=64 at 4 only), leff2 alternative shapes,
# and the erf-accurate f16 GELU (ab/erf) against the sigmoid form
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --no-other-modes --no-train-mode "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  gpu-sum', round(d['roofline']['gpu_ms_per_step_all_kernels'],3))"; }
{
for v in base lb4 lb4b; do if [ $v = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi; echo "hash $v $(python scripts/out_hash.py 2>/dev/null)"; done
for r in 1 2 3; do for v in base lb4 lb4b; do if [ $v = base ]; then unset UFORMER_HIP_LIB; else export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; fi; echo "$v run $r: $(b --kernels-json $O/k_$v.json)"; done; done
unset UFORMER_HIP_LIB
for r in 1 2; do echo "leff2 default run $r: $(b)"; echo "leff2 UF_LEFF2_VARIANT=a run $r: $(UF_LEFF2_VARIANT=a b)"; done
for r in 1 2 3; do echo "f16 sigmoid-GELU run $r: $(b --dtype f16)"; echo "f16 erf-GELU run $r: $(UFORMER_HIP_LIB=$R/ab/erf/libuformer_hip.so b --dtype f16)"; done
UFORMER_HIP_LIB=$R/ab/erf/libuformer_hip.so python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from oracle import uformer_oracle as O
from uformer_amd import model as um, spec
cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234); x = spec.synth_input(1, 256, 256, 1234)
ref = O.uformer_forward(x, sd, img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, dd_in=cfg.dd_in)
m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=torch.float16).eval(); m.load_state_dict(sd); m = m.cuda()
with torch.no_grad(): y = m(x.cuda()).cpu()
print("f16 erf-GELU max abs err vs oracle: %.3e" % (y - ref).abs().max().item())
PY
} 2>&1 | grep -v amdgpu.ids | tee $O/r03_occ_ab.txt
for v in base lb4; do echo "== $v"; python scripts/kernel_table.py $O/k_$v.json | grep -E "stage|enc0|enc1|dec3|all"; done | tee -a $O/r03_occ_ab.txt
