"""Per-k-tile timeline of gemm3 (GPU box; debug build with -DG3_TIMELINE loaded through SMTTS_LIB, tools/sessions/r03n.sh):
wave 0 of every workgroup stamps the shader clock (s_memtime, calibrated against s_memrealtime) at
   start | prologue DMAs issued | per k-tile: own DMA pieces landed (s_waitcnt) / barrier passed / next stage issued / MFMAs issued | k-loop end | epilogue end
The table gives, over all workgroups, the median time each k-tile spends WAITING for its own DMA, waiting at the BARRIER for the
other waves, ISSUING the next stage and in fragment reads + MFMAs — what a flag-based hand-off (no workgroup-wide barrier in the
k-loop) could at most recover is the barrier column."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine

eng = HipEngine(0, "f16")
lib = eng.lib
lib.smtts_debug_read_timeline.argtypes = [C.c_void_p, C.c_int]
SH = [("dit.qkvg 64x64 (round 2)", 600, 3840, 960, 0, 2, 1), ("dit.qkvg 64x128", 600, 3840, 960, 0, 0, 1), ("dit.out unsplit 64x64", 600, 960, 960, 0, 2, 1), ("dit.out unsplit 64x64, no bias", 600, 960, 960, 6, 2, 1),
      ("dit.ff2 unsplit 64x64", 600, 960, 2432, 0, 2, 1), ("codec s2.ff1 128x128", 24000, 2048, 512, 4, 1, 0), ("codec s1.ff1 128x128", 4800, 4096, 1024, 4, 1, 0)]
for name, M, N, K, epi, cfg, deep in SH:
    for dp in ((1, 0) if deep else (0,)):
        os.environ["SMTTS_GEMM_DEEP"] = str(dp)
        e2 = HipEngine(0, "f16")
        us = C.c_float()
        assert lib.smtts_debug_clear_timeline() == 0
        rc = lib.smtts_bench_gemm(e2.h, M, N, K, epi, 2, cfg, 1, 3, C.byref(us))
        assert rc == 0, lib.smtts_last_error(e2.h)
        buf = np.zeros(1024 * 160, np.uint64)
        assert lib.smtts_debug_read_timeline(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(1024, 160).astype(np.int64)
        live = t[:, 0] > 0
        t = t[live]
        # s_memtime counts the shader clock; the two s_memrealtime stamps (constant 100 MHz) around the kernel body calibrate it
        TICK_NS = float(np.median((t[:, 153] - t[:, 152]) * 10.0 / np.maximum(t[:, 151] - t[:, 0], 1)))
        nk = K // 64
        t0 = t[:, 0:1]
        rel = (t - t0) * TICK_NS / 1e3      # us
        own = np.stack([t[:, 2 + 4 * k] - (t[:, 5 + 4 * (k - 1)] if k else t[:, 1]) for k in range(nk)], 1) * TICK_NS / 1e3
        bar = np.stack([t[:, 3 + 4 * k] - t[:, 2 + 4 * k] for k in range(nk)], 1) * TICK_NS / 1e3
        iss = np.stack([t[:, 4 + 4 * k] - t[:, 3 + 4 * k] for k in range(nk)], 1) * TICK_NS / 1e3
        mma = np.stack([t[:, 5 + 4 * k] - t[:, 4 + 4 * k] for k in range(nk)], 1) * TICK_NS / 1e3
        epi_t = (t[:, 151] - t[:, 150]) * TICK_NS / 1e3
        tot = (t[:, 151] - t[:, 0]) * TICK_NS / 1e3
        med = lambda a: float(np.median(a))
        print(f"\n{name}  {M}x{N}x{K}  deep={dp}: {live.sum()} workgroups stamped (first 1024), launch {us.value:.1f} us by events; shader clock {1e3 / TICK_NS:.0f} MHz")
        print(f"  per workgroup (median): total {med(tot):.2f} us = prologue issue {med(rel[:, 1]):.2f} + k-loop {med(rel[:, 150] - rel[:, 1]):.2f} ({nk} k-tiles) + epilogue {med(epi_t):.2f}")
        print("  k-tile:        " + " ".join(f"{k:5d}" for k in range(nk)))
        for lab, a in (("wait own DMA", own), ("wait barrier", bar), ("issue next", iss), ("reads + MFMA", mma)):
            print(f"  {lab:13s}: " + " ".join(f"{med(a[:, k]):5.2f}" for k in range(nk)) + f"   sum {sum(med(a[:, k]) for k in range(nk)):.2f} us")
        e2.close()
