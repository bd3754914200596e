"""CPU: the C-ABI library loads and exports every symbol include/smalltts_hip.h declares,
and the ctypes signature table covers exactly that set. No compute calls (no GPU here)."""
import ctypes
import os

import pytest

from smalltts_amd import _lib


def _need_lib():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsmalltts_hip.so not built (run __graft_entry__.build())")


def test_header_and_signature_table_agree():
    assert sorted(_lib.SIGNATURES) == _lib.header_symbols()


def test_library_exports_every_declared_symbol():
    _need_lib()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/smalltts_hip.h but not exported"


def test_loader_sets_signatures_and_host_only_calls_work():
    _need_lib()
    lib = _lib.load()
    assert lib.smtts_version().startswith(b"smalltts-hip")
    a, s = ctypes.c_float(), ctypes.c_float()
    lib.smtts_alpha_sigma(ctypes.c_float(1.0), ctypes.byref(a), ctypes.byref(s))
    assert abs(a.value - 7.853981515e-6) < 1e-11 and s.value == 1.0


def test_schedule_matches_reference_kat():
    """smtts_alpha_sigma (host float64 math in C++) vs the reference's numpy values."""
    _need_lib()
    import numpy as np
    from tests.conftest import golden
    lib = _lib.load()
    k = golden("kat_schedule_rope.npz")
    for ts, ref in ((k["ts4"], k["alpha_sigma_4"]), (k["ts_dense"], k["alpha_sigma_dense"])):
        for t, (ra, rs) in zip(ts, ref):
            a, s = ctypes.c_float(), ctypes.c_float()
            lib.smtts_alpha_sigma(ctypes.c_float(float(t)), ctypes.byref(a), ctypes.byref(s))
            assert abs(a.value - float(ra)) <= 6e-8 * max(1.0, abs(float(ra)))
            assert abs(s.value - float(rs)) <= 6e-8 * max(1.0, abs(float(rs)))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsmalltts_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
