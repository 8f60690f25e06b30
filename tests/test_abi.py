"""The C-ABI library loads on a GPU-less box and exports exactly the symbols include/lmx.h declares (no compute)."""
import os
import re

import pytest

from lmx import _lib

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "lmx.h")


def _declared():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(lmx_[a-z0-9_]+)\s*\(", txt))


def test_header_and_bindings_agree():
    assert _declared() == set(_lib.SIGNATURES), (_declared() ^ set(_lib.SIGNATURES))


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build() first")
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.lmx_version() == 100


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_cpu_tensors_are_rejected():
    import torch

    from lmx import kernels as K

    a = torch.zeros((8, 8), dtype=torch.float16)
    with pytest.raises(K.LmxError):
        K.gemm(a, a)
