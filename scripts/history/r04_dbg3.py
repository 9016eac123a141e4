#!/usr/bin/env python3
"""round 4 debugging: variants of the LDS-staged stem under two streams; where the differing pixels are"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import os, sys, torch, hashlib
sys.path.insert(0, %r)
from uformer_amd import model as um, spec
cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234)
x = spec.synth_input(16, 256, 256, 1234).cuda()
m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=torch.bfloat16).eval(); m.load_state_dict(sd); m = m.cuda()
os.environ["UF_STREAMS_NOTE"] = ""
good = None; hs = {}
with torch.no_grad():
    ys = [m(x).clone() for _ in range(16)]
for y in ys:
    h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
    hs[h] = hs.get(h, 0) + 1
print(os.environ.get("TAG"), "distinct", len(hs), hs)
ref = [y for y in ys if hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12] == "b6846cb123c0"]
if ref and len(hs) > 1:
    for y in ys:
        d = (y - ref[0]).abs()
        if d.max() > 0:
            imgs = d.amax((1, 2, 3)).nonzero().flatten().tolist()
            i = imgs[0]
            idx = (d[i].amax(0) > 0).nonzero()
            print("   differs in images", imgs, "| image", i, "rows", idx[:, 0].min().item(), idx[:, 0].max().item(), "cols", idx[:, 1].min().item(), idx[:, 1].max().item(), "count", idx.shape[0], "max", d.max().item())
            break
""" % R
for tag, env in (("v2 default", {}), ("v2 bounds(256,1)", {"UFORMER_HIP_LIB": R + "/ab/ip1/libuformer_hip.so"}), ("v2 weights from global", {"UFORMER_HIP_LIB": R + "/ab/ip2/libuformer_hip.so"}),
                 ("v2 zero pad columns", {"UFORMER_HIP_LIB": R + "/ab/ip4/libuformer_hip.so"}), ("v2 + leff2 mc0-like never pw8", {"UF_LEFF2_VARIANT": "n"}), ("v2, attn 1 stream", {"UF_STREAMS": "1"}),
                 ("v2 UF_STREAMS=3", {"UF_STREAMS": "3"}), ("v1 UF_STREAMS=3", {"UF_STREAMS": "3", "UF_INPUT_PROJ_V1": "1"}), ("v1 UF_STREAMS=4", {"UF_STREAMS": "4", "UF_INPUT_PROJ_V1": "1"})):
    subprocess.call([sys.executable, "-c", code], env=dict(os.environ, TAG=tag, **env))
