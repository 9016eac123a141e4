"""Seeded probe vectors of the Uformer-B gradient fixture (tests/golden/make_golden_r2.py writes them, the tests re-derive
them): two signed random projections and one 4096-element gather per parameter.  A permuted, transposed or sign-flipped
gradient cannot pass these the way it passes an ``abs().sum()`` statistic."""
import hashlib

import torch


def proj_vector(name: str, k: int, shape) -> torch.Tensor:
    """r_k of a parameter: N(0,1) from a CPU generator seeded by sha256("gradproj:<name>:<k>")."""
    h = int.from_bytes(hashlib.sha256(f"gradproj:{name}:{k}".encode()).digest()[:6], "little")
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(h), dtype=torch.float32)


def gather_index(name: str, numel: int, n: int = 4096) -> torch.Tensor:
    h = int.from_bytes(hashlib.sha256(f"gradgather:{name}".encode()).digest()[:6], "little")
    return torch.randint(0, numel, (n,), generator=torch.Generator().manual_seed(h), dtype=torch.int64)
