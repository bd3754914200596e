"""Per-phase timeline of gemm4 (GPU box; debug build -DG4_TIMELINE loaded through SMTTS_LIB): waves 0 and 4 (one of either wave
row) of the first 256 workgroups stamp s_memtime four times per phase over k-tiles 4 and 5:
    phase start | fragment reads + unit DMA issued + vmcnt wait done | barrier 1 passed + reads landed | 8 MFMAs issued | (next start = barrier 2 passed)
Medians over workgroups, in shader-clock cycles (calibrated against s_memrealtime: 100 MHz)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, "f16")
lib = eng.lib
lib.smtts_debug_read_timeline4.argtypes = [C.c_void_p, C.c_int]
N = 256 * 2 * 2 * 16 + 8
for name, M, Nn, K, epi in [("sq4096", 4096, 4096, 4096, 0), ("s2.ff2", 24000, 512, 2048, 5), ("s1.ff1", 4800, 4096, 1024, 4)]:
    us = C.c_float()
    assert lib.smtts_debug_clear_timeline4() == 0
    assert lib.smtts_bench_gemm(eng.h, M, Nn, K, epi, 2, 7, 1, 3, C.byref(us)) == 0
    buf = np.zeros(N, np.uint64)
    assert lib.smtts_debug_read_timeline4(buf.ctypes.data, N) == 0
    cal = buf[-8:].astype(np.int64)
    mhz = (cal[3] - cal[1]) / max((cal[2] - cal[0]) * 0.01, 1e-9)   # shader ticks per us
    t = buf[:-8].astype(np.int64).reshape(256, 2, 32)     # [wg][wave row][k-tile 4: phases 0-3 x 4 stamps | k-tile 5: ...]
    live = (t[:, 0, 0] > 0) & (t[:, 1, 0] > 0)
    t = t[live]
    print(f"\n{name} {M}x{Nn}x{K}: {live.sum()} workgroups, launch {us.value:.1f} us by events, shader clock ~{mhz:.0f} MHz")
    for row in (0, 1):
        x = t[:, row, :]
        print(f"  wave row {row}:   phase   reads+issue+vmcnt   barrier1+lgkm   mfma issue   barrier2    total   (cycles, median)")
        for ph in range(8):
            b = 4 * ph
            nxt = x[:, b + 4] if ph < 7 else None
            seg = [x[:, b + 1] - x[:, b], x[:, b + 2] - x[:, b + 1], x[:, b + 3] - x[:, b + 2]]
            med = [float(np.median(s)) for s in seg]
            b2 = float(np.median(nxt - x[:, b + 3])) if nxt is not None else float("nan")
            tot = float(np.median(nxt - x[:, b])) if nxt is not None else float("nan")
            print(f"                  k{4 + ph // 4}.p{ph % 4}   {med[0]:12.0f}   {med[1]:12.0f}   {med[2]:10.0f}   {b2:8.0f}   {tot:7.0f}")
    d = t[:, 1, 0] - t[:, 0, 0]
    print(f"  wave row 1 starts k4.p0 {float(np.median(d)):.0f} cycles after wave row 0 (stagger)")
