"""CPU suite: the C-ABI library loads and exports every symbol include/guidegen_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "guidegen_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from jointimagegeneration_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in guidegen_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert lib.gg_version() >= 100


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the product path must fail loudly off-GPU."""
    import torch
    from jointimagegeneration_amd.unet import UNetModel
    from util import LDM_SMALL
    u = UNetModel(**LDM_SMALL).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        u(torch.zeros(1, 8, 16, 16), torch.tensor([1]))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "jointimagegeneration_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
