#!/bin/bash
# round 5, run 10 (final tree): (a) block / model fixtures with the halo-recompute LeFF selected; (b) the PCIe-inclusive rate of the headline step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(UF_LEFF3=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "leff or lewin or model" 2>&1 | tail -3) | tee $O/r05_run10_leff3_fixtures.txt
python - <<'PY' 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/r05_run10_pcie.txt
import time, torch, sys
sys.path.insert(0, ".")
from uformer_amd import model as um, spec
cfg = spec.arch_config("Uformer_B", img_size=256)
m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator, dd_in=cfg.dd_in,
               compute_dtype=torch.bfloat16).eval()
m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True); m = m.cuda()
xh = spec.synth_input(16, 256, 256, 1234).pin_memory()
yh = torch.empty_like(xh).pin_memory()
xd = xh.cuda()
def dev_only():
    return m(xd)
def with_pcie():
    y = m(xh.cuda(non_blocking=True)); yh.copy_(y, non_blocking=True)
with torch.no_grad():
    for name, fn in (("device-resident batch (the bench's value)", dev_only), ("pinned host batch in, pinned host batch out (PCIe both ways)", with_pcie)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{name}: {1e3 * dt:.3f} ms/step = {16 / dt:.1f} img/s")
PY
