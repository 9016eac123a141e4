#!/usr/bin/env python3
"""Where do the small ATen kernels of a training step come from?  One Uformer-B step (batch 8) under torch.profiler with Python stacks:
prints, per ATen op that launches a fill / copy / elementwise kernel, the call sites by number of calls."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from uformer_amd import losses as ul, model as um, optim as uo, spec


def main():
    cfg = spec.arch_config("Uformer_B", img_size=256)
    m = um.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads), modulator=cfg.modulator,
                   dd_in=cfg.dd_in, compute_dtype=torch.bfloat16)
    m.load_state_dict(spec.synth_state_dict(cfg, 1234), strict=True)
    m = m.cuda().train()
    opt = uo.AdamW(m.parameters(), lr=2e-4, weight_decay=0.02)
    crit = ul.CharbonnierLoss()
    x, t = spec.synth_input(8, 256, 256, 1).cuda(), spec.synth_input(8, 256, 256, 2).cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), t)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::div", "aten::add", "aten::cat", "aten::bernoulli", "aten::sub", "aten::neg", "aten::sum", "aten::_to_copy")
    sites = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    for ev in prof.events():
        if ev.name in want and ev.device_type == torch.autograd.DeviceType.CPU:
            stack = [s for s in (ev.stack or []) if "uformer_amd" in s or "scripts/" in s or "autograd" in s]
            site = stack[0] if stack else (ev.stack[0] if ev.stack else "?")
            sites[ev.name][site] += 1
            total[ev.name] += 1
    for name, n in total.most_common():
        print(f"{name}: {n} calls")
        for site, c in sites[name].most_common(6):
            print(f"    {c:5d}  {site}")
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and ("at::native" in ev.name or "rocclr" in ev.name):
            kern[ev.name[:90]] += 1
    print("ATen / runtime kernels of the step:")
    for k, c in kern.most_common(12):
        print(f"    {c:5d}  {k}")


if __name__ == "__main__":
    main()
