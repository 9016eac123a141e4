#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box does not have it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's ``model.py`` unmodified (a 3-symbol ``timm`` shim is put on
sys.path because ``timm`` is not installed: DropPath / to_2tuple / trunc_normal_,
model.py:4), feeds it seeded inputs and the deterministic weights of
``uformer_amd.spec.synth_state_dict`` and stores inputs/outputs as ``*.npz``.
The fixtures pin ``oracle/uformer_oracle.py`` (tests/test_oracle_golden.py) and are the
committed ground truth for the GPU parity tests.  Nothing here is imported by the product.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _install_timm_shim():
    d = tempfile.mkdtemp(prefix="timm_shim_")
    os.makedirs(os.path.join(d, "timm", "models"))
    open(os.path.join(d, "timm", "__init__.py"), "w").close()
    open(os.path.join(d, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "models", "layers.py"), "w") as f:
        f.write(
            "import torch, torch.nn as nn, collections.abc\n"
            "from itertools import repeat\n"
            "def to_2tuple(x):\n"
            "    return tuple(x) if isinstance(x, collections.abc.Iterable) and not isinstance(x, str) else tuple(repeat(x, 2))\n"
            "def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):\n"
            "    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)\n"
            "class DropPath(nn.Module):\n"
            "    def __init__(self, drop_prob=0.):\n"
            "        super().__init__(); self.drop_prob = drop_prob\n"
            "    def forward(self, x):\n"
            "        if self.drop_prob == 0. or not self.training: return x\n"
            "        keep = 1 - self.drop_prob\n"
            "        r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)\n"
            "        return x * r.div_(keep)\n")
    sys.path.insert(0, d)


_install_timm_shim()
sys.path.insert(0, REF)
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
import model as ref  # noqa: E402  (the reference's model.py)

from uformer_amd import spec  # noqa: E402


def g(seed):
    return torch.Generator().manual_seed(seed)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def randomize_(module, seed):
    """Fill every parameter of a reference sub-module with seeded non-trivial values."""
    gg = g(seed)
    with torch.no_grad():
        for n, p_ in module.named_parameters():
            r = torch.randn(p_.shape, generator=gg)
            if n.endswith("norm1.weight") or n.endswith("norm2.weight"):
                p_.copy_(1 + 0.1 * r)
            elif "relative_position_bias_table" in n:
                p_.copy_(0.3 * r)
            elif "modulator" in n:
                p_.copy_(0.5 * r)
            elif n.endswith("bias"):
                p_.copy_(0.1 * r)
            else:
                fan_in = p_[0].numel() if p_.ndim > 1 else 1
                p_.copy_(r * (1.0 / fan_in ** 0.5))


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    # ---------------- index-only ops (bit exact) ------------------------------------
    x = torch.arange(2 * 16 * 24 * 3, dtype=torch.int32).reshape(2, 16, 24, 3)
    wp = ref.window_partition(x, 8)
    wr = ref.window_reverse(wp, 8, 16, 24)
    assert torch.equal(wr, x)
    xs = torch.arange(2 * 16 * 16, dtype=torch.int64).reshape(2, 16, 16, 1)
    shifted = ref.window_partition(torch.roll(xs, shifts=(-4, -4), dims=(1, 2)), 8).reshape(-1)
    blk16 = ref.LeWinTransformerBlock(32, (16, 16), 1, win_size=8, shift_size=4, token_mlp="leff")
    blk32 = ref.LeWinTransformerBlock(32, (32, 32), 1, win_size=8, shift_size=4, token_mlp="leff")
    masks = {}
    for H, blk in ((16, blk16), (32, blk32)):
        # replay model.py:924-942 through the reference's own code path: capture the mask the
        # block hands to its attention module.
        cap = {}
        orig = blk.attn.forward

        def hook(xx, attn_kv=None, mask=None, _cap=cap, _orig=orig):
            _cap["mask"] = mask.clone()
            return _orig(xx, attn_kv, mask)

        blk.attn.forward = hook
        blk.eval()(torch.zeros(1, H * H, 32))
        masks[H] = cap["mask"]
    save("index_ops", x=x, partition=wp, shifted_index_16=shifted,
         shift_mask_16=masks[16], shift_mask_32=masks[32],
         rel_index=blk16.attn.relative_position_index)

    # ---------------- WindowAttention -------------------------------------------------
    attn = ref.WindowAttention(64, (8, 8), 2).eval()
    randomize_(attn, 11)
    xa = torch.randn(8, 64, 64, generator=g(12))
    save("window_attention", x=xa, mask=masks[16], y_nomask=attn(xa), y_mask=attn(xa, mask=masks[16]),
         **{"p." + k: v for k, v in attn.state_dict().items()})

    # ---------------- LeFF ---------------------------------------------------------------
    lf = ref.LeFF(16, 64).eval()
    randomize_(lf, 21)
    xl = torch.randn(2, 256, 16, generator=g(22))
    save("leff", x=xl, y=lf(xl), **{"p." + k: v for k, v in lf.state_dict().items()})

    # ---------------- LeWinTransformerBlock ----------------------------------------------
    for tag, C, heads, mod in (("a", 32, 1, False), ("b", 64, 2, True)):
        outs = {}
        b0 = ref.LeWinTransformerBlock(C, (16, 16), heads, win_size=8, shift_size=0, token_mlp="leff",
                                       modulator=mod).eval()
        randomize_(b0, 31 + C)
        b4 = ref.LeWinTransformerBlock(C, (16, 16), heads, win_size=8, shift_size=4, token_mlp="leff",
                                       modulator=mod).eval()
        b4.load_state_dict(b0.state_dict())
        xb = torch.randn(2, 256, C, generator=g(32 + C))
        outs["x"] = xb
        outs["y_shift0"] = b0(xb)
        outs["y_shift4"] = b4(xb)
        if tag == "b":   # user-mask path (model.py:914-921); B=1 because :942 only broadcasts for B=1
            um = (torch.rand(1, 1, 16, 16, generator=g(77)) > 0.4).float()
            outs["user_mask"] = um
            outs["y_shift4_usermask"] = b4(xb[:1], mask=um)
            outs["y_shift0_usermask"] = b0(xb[:1], mask=um)
        save("lewin_block_" + tag, heads=heads, **outs, **{"p." + k: v for k, v in b0.state_dict().items()})

    # ---------------- samplers / stem / head ---------------------------------------------
    dn = ref.Downsample(8, 16).eval(); randomize_(dn, 41)
    up = ref.Upsample(16, 8).eval(); randomize_(up, 42)
    ip = ref.InputProj(3, 8, 3, 1, act_layer=torch.nn.LeakyReLU).eval(); randomize_(ip, 43)
    op = ref.OutputProj(16, 3, 3, 1).eval(); randomize_(op, 44)
    xd = torch.randn(2, 256, 8, generator=g(45))
    xu = torch.randn(2, 64, 16, generator=g(46))
    xi = torch.rand(2, 3, 16, 16, generator=g(47)) - 0.3
    xo = torch.randn(2, 256, 16, generator=g(48))
    save("samplers", xd=xd, yd=dn(xd), xu=xu, yu=up(xu), xi=xi, yi=ip(xi), xo=xo, yo=op(xo),
         **{"dn." + k: v for k, v in dn.state_dict().items()},
         **{"up." + k: v for k, v in up.state_dict().items()},
         **{"ip." + k: v for k, v in ip.state_dict().items()},
         **{"op." + k: v for k, v in op.state_dict().items()})

    # ---------------- whole models -----------------------------------------------------------
    def run_model(tag, arch, img_size, B, HW, seed=1234, in_seed=1234):
        cfg = spec.arch_config(arch, img_size=img_size)
        sd = spec.synth_state_dict(cfg, seed)
        m = ref.Uformer(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths),
                        num_heads=list(cfg.num_heads), win_size=8, token_projection="linear",
                        token_mlp="leff", modulator=cfg.modulator, dd_in=cfg.dd_in).eval()
        ref_keys = list(m.state_dict().keys())
        assert ref_keys == [k for k, _, _ in spec.state_dict_spec(cfg)], "state_dict key order/layout drifted"
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
        m.load_state_dict(sd, strict=True)
        x = spec.synth_input(B, HW, HW, in_seed)
        y = m(x)
        save("model_" + tag, y=y, arch=arch, img_size=img_size, B=B, HW=HW, seed=seed, in_seed=in_seed,
             sd_sha256=sd_digest(sd), x_sha256=hashlib.sha256(x.numpy().tobytes()).hexdigest(),
             n_keys=len(ref_keys), n_params=sum(p_.numel() for p_ in m.parameters()))
        print("   |y-x| max %.4f mean %.4f" % ((y - x).abs().max(), (y - x).abs().mean()))

    run_model("tiny_128", "tiny", 128, 1, 128)              # BASELINE configs[0]
    run_model("tiny32_128", "tiny32", 128, 2, 128)          # smallest head_dim=32 model
    run_model("B_256", "Uformer_B", 256, 1, 256)            # configs[1] geometry, B=1
    run_model("B_ctor128_in256", "Uformer_B", 128, 1, 256)  # SURVEY Appendix A-1 trap (test scripts do this)


if __name__ == "__main__":
    main()
