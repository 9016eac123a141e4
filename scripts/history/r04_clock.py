#!/usr/bin/env python3
"""round 4 experiment: do the fused kernels run at the same shader clock inside the sustained model loop as in an isolated micro-benchmark?
(1) the attention-half kernel of dec1 / dec3 timed over 20 and over 3000 back-to-back launches; (2) a census (block life in ns and in
s_memtime cycles) taken during ONE forward in the middle of a sustained loop of forwards: the last kernels to run (dec3's leff2, then the
head) leave their entries; (3) rocm-smi clocks / power sampled while the loop runs."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uformer_amd import _lib, model as um, spec
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ubench

lib = _lib.load()
for (B, H, C, heads) in ((16, 64, 256, 8), (16, 256, 64, 2), (16, 32, 512, 16)):
    blk = um.LeWinTransformerBlock(C, (H, H), heads, win_size=8, shift_size=4, modulator=True).cuda().eval()
    bp = blk._pack(torch.bfloat16)
    M = B * H * H
    x = torch.randn(M, C, device="cuda")
    nbytes = lib.uf_block_workspace_bytes(M, C, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    fa = lambda: lib.uf_lewin_attn_fwd(bp, x.data_ptr(), C, B, H, H, C, None, 0, 1, ws.data_ptr(), nbytes, st)
    fl = lambda: lib.uf_leff_fwd(bp, x.data_ptr(), C, B, H, H, C, 1, ws.data_ptr(), nbytes, st)
    def both():
        fa(); fl()
    for n in (20, 3000):
        print(f"M={M} C={C}: attn_half x{n}: {ubench.timeit(fa, n=n, warm=3):7.1f} us   leff2 x{n}: {ubench.timeit(fl, n=n, warm=3):7.1f} us   alternating x{n}: {ubench.timeit(both, n=n, warm=3):7.1f} us per pair", flush=True)
    del x, ws

cfg = spec.arch_config("Uformer_B", img_size=256); sd = spec.synth_state_dict(cfg, 1234)
m = um.Uformer(img_size=256, embed_dim=32, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=True, compute_dtype=torch.bfloat16).eval()
m.load_state_dict(sd); m = m.cuda()
x = spec.synth_input(16, 256, 256, 1234).cuda()
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().replace("\n", " | ")[:400])
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.2)
th = threading.Thread(target=sampler); th.start()
tb = torch.zeros(ubench.TBUF_ELEMS, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): m(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"sustained loop: {(t1 - t0) / 300 * 1e3:.3f} ms per forward (300 forwards)")
    for _ in range(100): m(x)
    lib.uf_debug_set_tbuf(tb.data_ptr())
    m(x)
    torch.cuda.synchronize()
    lib.uf_debug_set_tbuf(None)
    for _ in range(50): m(x)
    torch.cuda.synchronize()
stop = True; th.join()
ubench.census_report("census entries left by the last kernels of a forward inside the sustained loop (UF_STREAMS default)", tb, 16384)
print("rocm-smi samples during the loop:")
for s in samples[:3] + samples[len(samples) // 2: len(samples) // 2 + 3] + samples[-2:]:
    print("  ", s)
