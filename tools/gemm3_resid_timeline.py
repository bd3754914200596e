"""Where a launch of the DiT's N = 960 projections spends its time.  GPU box, lab build
    make -C smalltts_amd/csrc LAB=1 BUILD=build_tl LIB=../libsmalltts_hip_tl.so EXTRA="-DG3_TIMELINE -DG3_TL_NO_K"
loaded through SMTTS_LIB (no per-k-tile stamps: one s_memtime per k-tile doubles the loop of this one-wave-per-SIMD tile; the loop's
own cost comes from tools/gemm_kloop_probe.py on the un-instrumented library): wave 0 of every workgroup stamps the shader clock (s_memtime, calibrated per
workgroup against s_memrealtime) at
   start | ring prologue issued | first k-tile landed | k-loop end | mask bytes landed | residual + vectors landed | stores issued | end
and the 100 MHz real-time counter at start / end, which places every workgroup on ONE time axis: first workgroup's start -> last
workgroup's end is the kernel's body; the HIP-event time of the launch minus that is dispatch + completion.
    python tools/gemm3_resid_timeline.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine

eng = HipEngine(0, "f16")
lib = eng.lib
lib.smtts_debug_read_timeline_resid.argtypes = [C.c_void_p, C.c_int]
# name, M, N, K, epi, cfg (2 = 64x64, 8 = 64x32, 9 = 32x64), ring ("SMTTS_GEMM_DEEP" value)
SH = [("out-proj resid_gate + mask, 64x64", 600, 960, 960, 8, 2), ("out-proj resid_ln + mask, 64x64", 600, 960, 960, 7, 2),
      ("FF2 resid_gate, 64x64", 600, 960, 2432, 3, 2), ("FF2 resid_ln, 64x64", 600, 960, 2432, 9, 2)]
for extra in os.environ.get("TL_CFGS", "").split():
    c = int(extra)
    SH += [(f"out-proj resid_ln + mask, cfg {c}", 600, 960, 960, 7, c), (f"FF2 resid_ln, cfg {c}", 600, 960, 2432, 9, c)]
for name, M, N, K, epi, cfg in SH:
    for dp in os.environ.get("TL_DEEP", "1").split():
        os.environ["SMTTS_GEMM_DEEP"] = dp
        e2 = HipEngine(0, "f16")
        us = C.c_float()
        best = 1e9
        for _ in range(3):   # the un-stamped launch time: 50 launches back to back
            assert lib.smtts_bench_gemm(e2.h, M, N, K, epi, 2, cfg, 50, 3, C.byref(us)) == 0, lib.smtts_last_error(e2.h)
            best = min(best, us.value)
        assert lib.smtts_debug_clear_timeline_resid() == 0
        assert lib.smtts_bench_gemm(e2.h, M, N, K, epi, 2, cfg, 1, 3, C.byref(us)) == 0, lib.smtts_last_error(e2.h)
        buf = np.zeros(1024 * 160, np.uint64)
        assert lib.smtts_debug_read_timeline_resid(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(1024, 160).astype(np.int64)
        t = t[t[:, 0] > 0]
        tick = np.median((t[:, 153] - t[:, 152]) * 10.0 / np.maximum(t[:, 151] - t[:, 0], 1))   # ns per shader tick
        us_of = lambda a, b: (t[:, a] - t[:, b]) * tick / 1e3
        med = lambda a: float(np.median(a))
        start = (t[:, 152] - t[:, 152].min()) * 0.01     # us on the common axis
        end = (t[:, 153] - t[:, 152].min()) * 0.01
        nk = K // 64
        print(f"\n{name}  {M}x{N}x{K}  ring depth switch {dp}: {len(t)} workgroups, launch {best:.1f} us (events, back to back), "
              f"single stamped launch {us.value:.1f} us")
        print(f"  common axis: workgroup starts 0 .. {start.max():.2f} us (median {med(start):.2f}); last end {end.max():.2f} us; "
              f"body (first start -> last end) {end.max():.2f} us")
        print(f"  per workgroup (median / max): total {med(us_of(151, 0)):.2f} / {us_of(151, 0).max():.2f} us = prologue issue {med(us_of(1, 0)):.2f}"
              f" | k-loop ({nk} k-tiles) +{med(us_of(150, 1)):.2f}"
              f" | mask bytes +{med(us_of(154, 150)):.2f} | residual + vectors +{med(us_of(155, 154)):.2f} | stores issued +{med(us_of(156, 155)):.2f}"
              f" | {'partials + ' if epi in (7, 9) else ''}drain +{med(us_of(151, 156)):.2f}")
        e2.close()
