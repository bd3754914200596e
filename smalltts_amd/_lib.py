"""ctypes loader for libsmalltts_hip.so (the C ABI declared in include/smalltts_hip.h).

The product path has no CPU fallback: if the shared library is missing or cannot be
loaded, importing an operator raises immediately."""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMTTS_LIB") or os.path.join(_HERE, "libsmalltts_hip.so")  # SMTTS_LIB: A/B builds (csrc/Makefile)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "smalltts_hip.h")

_lib = None
ABI_VERSION = 5   # include/smalltts_hip.h SMTTS_ABI_VERSION

vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
cstr = C.c_char_p

# name -> (restype, argtypes); kept in sync with include/smalltts_hip.h (tests/test_cabi.py checks it)
SIGNATURES: Dict[str, Tuple[object, List[object]]] = {
    "smtts_create": (i32, [i32, C.POINTER(vp)]),
    "smtts_destroy": (i32, [vp]),
    "smtts_last_error": (cstr, [vp]),
    "smtts_version": (cstr, []),
    "smtts_abi_version": (i32, []),
    "smtts_get_saturations": (i32, [vp, C.POINTER(C.c_uint32), i32, i32]),
    "smtts_range_report": (cstr, [vp]),
    "smtts_range_worst_bound": (f32, [vp]),
    "smtts_set_tensor": (i32, [vp, cstr, vp, C.POINTER(i64), i32, i32]),
    "smtts_synth_tensor": (i32, [vp, cstr, C.POINTER(i64), i32, u64, f32, f32]),
    "smtts_get_tensor": (i32, [vp, cstr, vp, i64]),
    "smtts_set_codec_spec": (i32, [vp, i32, i32, i32, i32, f32, C.POINTER(i32), i32, C.POINTER(i32)]),
    "smtts_finalize": (i32, [vp]),
    "smtts_set_precision": (i32, [vp, i32]),
    "smtts_get_precision": (i32, [vp]),
    "smtts_default_precision": (i32, []),
    "smtts_set_site_precision": (i32, [vp, i32, i32]),
    "smtts_has_part": (i32, [vp, i32]),
    "smtts_cond_workspace_bytes": (sz, [vp, i32, i32, i32]),
    "smtts_cond_encode": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp, vp]),
    "smtts_denoise_workspace_bytes": (sz, [vp, i32, i32, i32, i32]),
    "smtts_denoise_step": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz]),
    "smtts_sample_workspace_bytes": (sz, [vp, i32, i32, i32, i32, i32, i32]),
    "smtts_sample": (i32, [vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, u64,
                           vp, vp, vp, sz]),
    "smtts_codec_hop": (i32, [vp]),
    "smtts_decode_workspace_bytes": (sz, [vp, i32, i32]),
    "smtts_codec_decode": (i32, [vp, vp, vp, i32, i32, vp, vp, sz]),
    "smtts_encode_workspace_bytes": (sz, [vp, i32, i32]),
    "smtts_codec_encode": (i32, [vp, vp, vp, i32, i32, vp, vp, sz]),
    "smtts_set_dual_stream": (i32, [vp, i32]),
    "smtts_set_tuning": (i32, [vp, i32]),
    "smtts_randn": (i32, [vp, vp, vp, i64, u64, u64]),
    "smtts_resample_poly": (i32, [vp, vp, vp, i32, i64, vp, i32, i32, i32, i32, vp, i64]),
    "smtts_pcm16": (i32, [vp, vp, vp, i64, vp]),
    "smtts_alpha_sigma": (None, [f32, C.POINTER(f32), C.POINTER(f32)]),
    "smtts_profile_enable": (i32, [vp, i32]),
    "smtts_profile_report": (i32, [vp, C.c_char_p, sz]),
    "smtts_bench_gemm": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(f32)]),
    "smtts_test_gemm3": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "smtts_test_set_fused_ffn": (i32, [vp, i32]),
    "smtts_test_set_ln_fold": (i32, [vp, i32]),
    "smtts_test_gemm": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32]),
    "smtts_test_swiglu": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "smtts_test_set_attention_mfma": (i32, [vp, i32]),
    "smtts_test_attention_mfma": (i32, [vp, vp, vp, vp, vp, f32, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32,
                                        i32, i32, vp]),
    "smtts_test_attention": (i32, [vp, vp, vp, vp, vp, f32, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32,
                                   i32, i32, vp]),
}


def header_symbols() -> List[str]:
    """Entry points declared in include/smalltts_hip.h."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(smtts_[a-z_0-9]+)\s*\(", txt)))


def load():
    """Load the HIP library; raises if it was not built (python __graft_entry__.py / make -C smalltts_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C smalltts_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm first: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so), and this library must bind to THAT one —
    # two HIP runtimes in one process do not both see the GPU.  Loaded the other way round (this library, then torch: what
    # `python __graft_entry__.py smoke` does when build() runs in front of smoke()) smtts_create fails with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.smtts_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.smtts_abi_version()}, this host side was written for {ABI_VERSION} "
                           "(include/smalltts_hip.h SMTTS_ABI_VERSION): rebuild with `make -C smalltts_amd/csrc`")
    _lib = lib
    return lib
