"""The C-ABI shared library loads and exports every symbol include/*.h declares.

No compute calls here (there is no GPU in the build container); scoring entry
points must fail loudly with S3A_ENODEV instead of falling back to a CPU path.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from cmusphinx_amd import lib


def declared_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            src = open(os.path.join(inc, fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            syms.update(re.findall(r"\b(s3a_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_loads_and_exports_every_declared_symbol():
    L = lib.load()
    assert lib.MISSING == [], f"binding names symbols the library lacks: {lib.MISSING}"
    decl = declared_symbols()
    assert len(decl) > 50
    for name in sorted(decl):
        assert hasattr(L, name), f"{name} is declared in include/ but not exported"
    # and the binding covers the whole header
    assert decl == set(lib._SIGS), (decl ^ set(lib._SIGS))


def test_version_and_error_strings():
    L = lib.load()
    assert b"gfx950" in L.s3a_version()
    assert isinstance(L.s3a_last_error(), bytes)


def test_no_cpu_fallback_without_a_device():
    """Without a GPU the model constructor must refuse (S3A_ENODEV), not fall back."""
    if lib.device_count() > 0:
        pytest.skip("a GPU is present; the refusal path is exercised on the CPU box")
    lm = lib.LogMath(1.0003)
    with pytest.raises(lib.S3AError, match="no HIP device"):
        lib.MgauModel.init_arrays(np.ones((2, 2, 4), np.float32), np.ones((2, 2, 4), np.float32),
                                  np.ones((2, 2), np.float32), lm)


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under cmusphinx_amd/ or include/ may use it."""
    bad = []
    for base in ("cmusphinx_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"s3o_|libs3oracle|oracle_lib|oracle/", txt):
                        bad.append(os.path.join(dp, fn))
    assert bad == []
    out = os.popen(f"ldd {lib.LIB_PATH}").read()
    assert "s3oracle" not in out and "s3ref" not in out
