"""The C-ABI library loads without a GPU and exports every function include/ldpc_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ldpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_header_symbols():
    from ldpc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ldpc_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ldpc_amd/_lib.py SYMBOLS out of sync with the header"


def test_version_and_error_strings():
    from ldpc_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.ldpc_hip_version()
    assert isinstance(lib.ldpc_hip_last_error(), bytes)


def test_invalid_arguments_are_reported_not_fatal():
    from ldpc_amd import _lib
    lib = _lib.load()
    out = ctypes.c_void_p()
    assert lib.ldpc_hip_bp_create(None, ctypes.byref(out)) == -1
    assert b"null" in lib.ldpc_hip_last_error()
    assert lib.ldpc_hip_bp_set_params(None, 1, 0, 1.0) == -1
    assert lib.ldpc_hip_bp_decode_batch(None, None, 1, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from ldpc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libldpc_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
