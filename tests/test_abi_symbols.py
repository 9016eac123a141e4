"""CPU: the C-ABI library builds/loads and exports every symbol include/uformer_hip.h declares,
and the ctypes binding lists exactly those symbols.  No compute (no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "uformer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from uformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/uformer_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    bound = _lib.load()
    assert bound.uf_version() == 1


def test_errors_cross_the_abi_as_codes():
    """Argument validation happens before any launch, so it is checkable without a GPU."""
    from uformer_amd import _lib
    lib = _lib.load()
    rc = lib.uf_linear_fwd(None, None, None, None, 64, 64, 64, 0, 0, None)
    assert rc == -6 and "null" in _lib.last_error()
    rc = lib.uf_window_partition(16, 32, 1, 12, 16, 4, 0, 4, None)
    assert rc == -1 and "H=12" in _lib.last_error()
    rc = lib.uf_window_attention_fwd(16, 16, 16, 16, None, 0, 16, 4, 2, 24, 16, 16, 0, 1, None)
    assert rc == -2 and "head_dim" in _lib.last_error()
    assert lib.uf_block_workspace_bytes(4096, 64, 1) == 4096 * 64 * 2 * 9
    with pytest.raises(_lib.UformerHipError):
        _lib.check(rc, "x")


def test_cpu_tensors_raise_no_fallback():
    import torch
    from uformer_amd import model, ops
    from uformer_amd._lib import UformerHipError
    with pytest.raises(UformerHipError):
        ops.window_partition(torch.zeros(1, 8, 8, 4), 8, 0)
    m = model.get_arch("Uformer_T", 128).eval()
    with pytest.raises(UformerHipError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 128, 128))
    with pytest.raises(UformerHipError, match="no CPU fallback"):      # the training path is HIP-only as well
        m.train()(torch.zeros(1, 3, 128, 128))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "uformer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "uformer_oracle" not in txt, f
