"""Tile-configuration sweep of gemm3 on the DiT / teacher product shapes at fp16 (run on the GPU box; SMTTS_GEMM_DEEP=0|1):
    python tools/gemm_cfg_sweep.py      cfg -1 = what gemm3_pick_cfg chooses"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, "f16")
SH = [("dit.qkvg", 600, 3840, 960, 0), ("dit.ff1", 600, 4800, 960, 2), ("dit.out.unsplit", 600, 960, 960, 0), ("dit.ff2.unsplit", 600, 960, 2432, 0),
      ("qkvg.teacher", 1800, 3840, 960, 0), ("out.teacher", 1800, 960, 960, 0), ("ff2.teacher", 1800, 960, 2432, 0),
      ("ff1.teacher", 1800, 4800, 960, 2), ("qkvg.b4", 300, 3840, 960, 0), ("qkvg.b16", 1200, 3840, 960, 0), ("enc.qkvg", 240, 2048, 512, 0)]
for name, M, N, K, epi in SH:
    for cfg in (-1, 0, 1, 2, 5, 6):
        if epi == 2 and cfg != -1: continue   # (SwiGLU pairs: the launcher picks 128x128 or 160x128 itself)
        us = C.c_float()
        rc = eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, 2, cfg, 50, 3, C.byref(us))
        print(f"{name:18s} cfg {cfg:2d} deep={os.environ.get('SMTTS_GEMM_DEEP','-')} : " + (f"{us.value:7.1f} us  {2.0*M*N*K/us.value/1e6:7.1f} TF/s" if not rc else "error " + eng.lib.smtts_last_error(eng.h).decode()))
