"""Tests of code written when round 1 had no GPU time left: NOT under the ``gpu`` marker yet (the round-end GPU run selects
``-m gpu``), skipped without a GPU.  Run them first thing in the next round (``pytest tests/test_gpu_next.py``), then move
them under the gpu marker."""
import pytest
import torch

pytestmark = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu_fwd(dtype):
    from oracle import uformer_oracle as O
    from uformer_amd import ops
    a = (torch.randn(5, 33, 64) * 2).to(dtype)
    ref = O.gelu_erf(a.float())
    got = ops.gelu(a.cuda()).float().cpu()
    assert (got - ref).abs().max() < (1e-6 if dtype == torch.float32 else 2.5e-2)


def test_training_forward_with_separate_gelu_matches(monkeypatch):
    """UF_TRAIN_SEPARATE_GELU path of uformer_amd/train.py against the default path (same block, same gradients)."""
    import numpy as np
    import os
    from uformer_amd import train
    gd = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "grad_lewin_block.npz")))
    t = lambda a: torch.from_numpy(np.asarray(a))                           # noqa: E731
    p = {k[2:]: t(v).cuda() for k, v in gd.items() if k.startswith("p.")}
    monkeypatch.setattr(train, "_SEPARATE_GELU", True)
    y, dx, grads = train.lewin_block_forward_backward(t(gd["x"]).cuda(), p, "", int(gd["heads"]), 4, t(gd["gy"]).cuda(), torch.float32)
    rel = lambda a, b: (a.float().cpu() - b).abs().max().item() / b.abs().max().item()    # noqa: E731
    assert rel(y, t(gd["y"])) < 1e-3 and rel(dx, t(gd["dx"])) < 1e-3
    for k, r in ((k[2:], t(v)) for k, v in gd.items() if k.startswith("g.")):
        assert rel(grads[k], r) < 1e-3, k
