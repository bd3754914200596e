"""Tile-configuration sweep of gemm3 on the codec's many-round products at fp16 (GPU box):  python tools/gemm_codec_sweep.py
cfg -1 = what gemm3_pick_cfg chooses; 1 128x128 (8 waves), 6 128x128 (4 waves 64x64), 5 160x128.
epi 4 = GELU hidden written as one fp16 array (first FFN product), 5 = LayerScale residual into fp32 x (second), 0 = fp32 store."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, "f16")
SH = [("s2.ff1", 24000, 2048, 512, 4), ("s2.ff2", 24000, 512, 2048, 5), ("s1.ff1", 4800, 4096, 1024, 4), ("s1.ff2", 4800, 1024, 4096, 5),
      ("s0.ff1", 600, 8192, 2048, 4), ("s0.ff2", 600, 2048, 8192, 5),
      ("up.s2", 4800, 2560, 2048, 0), ("up.s3", 24000, 1280, 1024, 0), ("up.s4", 120000, 512, 512, 0),
      ("s3.ff1", 120000, 1024, 256, 4), ("s3.ff2", 120000, 256, 1024, 5),
      ("teacher.out", 1800, 960, 960, 3), ("teacher.ff2", 1800, 960, 2432, 3)]
for name, M, N, K, epi in SH:
    for cfg in (-1, 1, 6, 5):
        us = C.c_float()
        rc = eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, 2, cfg, 30, 3, C.byref(us))
        print(f"{name:12s} {M:6d}x{N:5d}x{K:5d} epi {epi} cfg {cfg:2d}: " + (f"{us.value:7.1f} us  {2.0*M*N*K/us.value/1e6:7.1f} TF/s" if not rc else "error " + eng.lib.smtts_last_error(eng.h).decode()), flush=True)
