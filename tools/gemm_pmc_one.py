"""One GEMM shape / config a few times, for rocprofv3 --pmc passes:  python tools/gemm_pmc_one.py M N K epi cfg [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
M, N, K, epi, cfg = (int(v) for v in sys.argv[1:6])
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
eng = HipEngine(0, "f16")
us = C.c_float()
assert eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, 2, cfg, it, 3, C.byref(us)) == 0
print(f"{M}x{N}x{K} epi {epi} cfg {cfg}: {us.value:.1f} us")
