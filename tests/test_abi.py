"""CPU: the C-ABI library loads and exports every symbol include/daliid.h declares (no compute)."""
import os
import re

import pytest

from conftest import ROOT
from daliid_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "daliid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dali_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    declared = _declared()
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), "libdaliid_hip.so does not export %s" % name
    assert sorted(_lib.exported_symbols()) == declared, "ctypes table and header disagree"
    assert L.dali_version() >= 100


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DaliError):
        _lib.ctx()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under daliid_amd/ may import it."""
    pkg = os.path.join(ROOT, "daliid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
