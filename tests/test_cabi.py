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


def test_one_precision_default_for_the_c_abi_and_the_python_host_side():
    """A handle from smtts_create starts at the same preset the Python engine asks for ("f16 mixed" = 2), and the header says so."""
    _need_lib()
    from smalltts_amd import engine as weights
    lib = _lib.load()
    assert lib.smtts_default_precision() == 2
    assert weights.PRECISION[weights.DEFAULT_PRECISION] == 2
    with open(_lib.HEADER_PATH) as f:
        txt = f.read()
    assert "THE DEFAULT of a new handle" in txt and "single-stream" not in txt


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


def test_null_handle_is_an_error_not_a_crash():
    """Every entry point that takes a handle returns an error code for NULL (size queries: 0) and leaves a message
    for smtts_last_error(NULL); none of them touches HIP before the check, so this runs without a GPU."""
    _need_lib()
    lib = _lib.load()
    skip = {"smtts_create", "smtts_destroy", "smtts_last_error", "smtts_version", "smtts_alpha_sigma", "smtts_default_precision",
            "smtts_abi_version"}
    for name, (res, args) in _lib.SIGNATURES.items():
        if name in skip:
            continue
        assert args and args[0] is _lib.vp, name
        call = [None] + [(ctypes.c_float(0) if a is _lib.f32 else None if a in (_lib.vp, _lib.cstr) or hasattr(a, "contents")
                          or a is ctypes.c_char_p else 0) for a in args[1:]]
        rc = getattr(lib, name)(*call)
        if name == "smtts_range_report":
            assert rc == b"", name
        elif res is _lib.sz or name in ("smtts_has_part", "smtts_codec_hop", "smtts_range_worst_bound"):
            assert rc == 0, name
        else:
            assert rc == 1, name
        assert name.encode() in lib.smtts_last_error(None), name
    assert lib.smtts_destroy(None) == 0   # like free(NULL)


def test_abi_version_matches_the_header_and_the_host_side():
    """SMTTS_ABI_VERSION of include/smalltts_hip.h == smtts_abi_version() of the built library == what smalltts_amd/_lib.py was
    written for (a stale .so must fail at load, not at the first mismatching call)."""
    import re
    _need_lib()
    lib = _lib.load()
    with open(_lib.HEADER_PATH) as f:
        m = re.search(r"#define\s+SMTTS_ABI_VERSION\s+(\d+)", f.read())
    assert m and int(m.group(1)) == lib.smtts_abi_version() == _lib.ABI_VERSION
    assert b"0.5" in lib.smtts_version()
