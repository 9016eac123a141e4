#!/bin/bash
# which switch makes the restored batch irreproducible / batch-dependent?  out_hash (batch 16) three times per setting + batch-1 hash of image 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
h() { for i in 1 2 3; do python scripts/out_hash.py 2>/dev/null; done | sort | uniq -c | tr '\n' ';'; }
{
echo "default:            $(h)"
echo "leff2 never pw8:    $(UF_LEFF2_VARIANT=n h)"
echo "leff2 always pw8:   $(UF_LEFF2_VARIANT=p h)"
echo "input_proj v1:      $(UF_INPUT_PROJ_V1=1 h)"
echo "output_proj v1:     $(UF_OUTPUT_PROJ_V1=1 h)"
echo "stem+head v1, n:    $(UF_INPUT_PROJ_V1=1 UF_OUTPUT_PROJ_V1=1 UF_LEFF2_VARIANT=n h)"
python - <<'PY'
import os, sys, hashlib, torch
sys.path.insert(0, '.')
from uformer_amd import model as um, spec
cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234)
x = spec.synth_input(16, 256, 256, 1234).cuda()
m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=torch.bfloat16).eval(); m.load_state_dict(sd); m = m.cuda()
with torch.no_grad():
    y = m(x); y2 = m(x)
    print("repeat equal:", bool(torch.equal(y, y2)), "max diff", (y - y2).abs().max().item())
    for i in (0, 7, 15):
        yi = m(x[i:i + 1])
        d = (yi - y[i:i + 1]).abs()
        print(f"image {i}: batch-1 vs batch-16 max diff {d.max().item():.3e}, differing pixels {int((d > 0).sum())}")
        if d.max() > 0:
            idx = (d[0].amax(0) > 0).nonzero()
            print("   rows", idx[:, 0].min().item(), idx[:, 0].max().item(), "cols", idx[:, 1].min().item(), idx[:, 1].max().item())
PY
} 2>&1 | grep -v amdgpu.ids | tee $O/r04_bisect.txt
