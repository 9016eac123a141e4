#!/usr/bin/env python3
"""Round 4, stale-row diagnosis 7: (a) a faster FIRST-form stem (8 pixels per thread, no LDS), (b) the LDS-staged stem with an empty kernel between it and
its consumer, (c) the LDS-staged stem with the first attn_block in its first form -- ten forwards each over a refilled workspace, two half-batch streams."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import os, sys, torch, hashlib
sys.path.insert(0, %r)
from uformer_amd import model as um, spec
cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234)
x = spec.synth_input(16, 256, 256, 1234).cuda()
m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=torch.bfloat16).eval(); m.load_state_dict(sd); m = m.cuda()
hs = {}
with torch.no_grad():
    y = m(x)
    for poison in (0xFF, 0x00, 0xFF, 0x7F, 0xFF, 0x00, 0xFF, 0xFF, 0x00, 0xFF):
        for ws in m._ws.values():
            ws.fill_(poison)
        torch.cuda.synchronize()
        h = hashlib.sha256(m(x).cpu().numpy().tobytes()).hexdigest()[:12]
        hs[h] = hs.get(h, 0) + 1
print(os.environ.get("TAG"), hs, flush=True)
""" % R
V2 = {"UF_INPUT_PROJ_V2": "1"}
for tag, env in (("first form, 4 pixels per thread", {}), ("first form, 8 pixels per thread", {"UF_INPUT_PROJ_PX": "8"}), ("first form, 8 pixels per thread, again", {"UF_INPUT_PROJ_PX": "8"}),
                 ("LDS-staged", V2), ("LDS-staged + empty kernel behind it", dict(V2, UF_IP2_FENCE="1")), ("LDS-staged + empty kernel behind it, again", dict(V2, UF_IP2_FENCE="1")),
                 ("LDS-staged, attn_block first form everywhere", dict(V2, UF_ATTN_LR="0")), ("LDS-staged, leff2 one tile per workgroup, never 8 producers", dict(V2, UF_LEFF2_PERSIST="0", UF_LEFF2_VARIANT="n"))):
    subprocess.call([sys.executable, "-c", code], env=dict(os.environ, TAG=tag, **env))
