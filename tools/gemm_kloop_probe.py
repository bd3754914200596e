"""Is the k-loop of the 64x64 tile (the DiT's N = 960 projections) bound by the workgroup's own pipeline or by the memory system?
Launch time (HIP events, 200 launches back to back) against K for ONE workgroup, one XCD's worth, and the full 150-workgroup grid:
the slope is the time per k-tile, the intercept the fixed cost.    python tools/gemm_kloop_probe.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine

eng = HipEngine(0, "f16")
lib = eng.lib
us = C.c_float()
print("M x N (workgroups of 64x64)      K=256    K=960   K=2432   K=4864   -> us per k-tile (slope 960 -> 4864), fixed (K -> 0)")
for dp in os.environ.get("TL_DEEP", "1 2").split():
    os.environ["SMTTS_GEMM_DEEP"] = dp
    e2 = HipEngine(0, "f16")
    for M, N in ((64, 64), (64, 512), (320, 512), (600, 960), (1200, 960)):
        row = []
        for K in (256, 960, 2432, 4864):
            best = 1e9
            for _ in range(3):
                assert lib.smtts_bench_gemm(e2.h, M, N, K, int(os.environ.get("TL_EPI", "3")), 2, 2, 200, 3, C.byref(us)) == 0
                best = min(best, us.value)
            row.append(best)
        slope = (row[3] - row[1]) / ((4864 - 960) / 64)
        print(f"ring {dp}: {M:5d} x {N:4d} ({((M + 63) // 64) * (N // 64):4d} wg)   " + "  ".join(f"{v:7.1f}" for v in row) + f"   {slope:.3f}  {row[1] - slope * 15:.1f}")
    e2.close()
