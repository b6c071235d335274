"""CPU: the C-ABI library loads and exports every symbol include/daliid.h declares (no compute)."""
import os
import re

import pytest

from conftest import ROOT
from daliid_amd import _lib


def _declared(header="daliid.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
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


def test_library_exports_exactly_what_the_headers_declare():
    """`nm -D` of the shared library against the two headers: the drop-in surface (daliid.h) plus the diagnostic entry points
    (daliid_debug.h, bound by scripts/ only).  An export that no header names, or a declaration without a definition, fails here."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TW" and ln.split()[-1].startswith("dali_")})
    surface, debug = _declared(), _declared("daliid_debug.h")
    assert debug and all(n.startswith("dali_debug_") for n in debug)
    assert not set(surface) & set(debug)
    assert exported == sorted(surface + debug), (sorted(set(exported) - set(surface + debug)), sorted(set(surface + debug) - set(exported)))
    assert not [n for n in _lib.exported_symbols() if n.startswith("dali_debug_")], "the product binds a diagnostic entry point"


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
