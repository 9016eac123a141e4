#!/usr/bin/env python3
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
print(os.environ.get("TAG"), hs)
""" % R
for tag, env in (("v2 dynamic LDS", {"UFORMER_HIP_LIB": R + "/ab/ip34/libuformer_hip.so"}), ("v2 slower (sleep)", {"UFORMER_HIP_LIB": R + "/ab/ip66/libuformer_hip.so"}),
                 ("v2 default again", {"UFORMER_HIP_LIB": R + "/ab/ip2/libuformer_hip.so"}), ("v2 default, HIP_LAUNCH_BLOCKING-free but GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}),
                 ("v2 + threadfence, run 2", {"UFORMER_HIP_LIB": R + "/ab/ip10/libuformer_hip.so"})):
    subprocess.call([sys.executable, "-c", code], env=dict(os.environ, TAG=tag, **env))
