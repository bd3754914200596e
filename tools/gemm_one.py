#!/usr/bin/env python
"""Run ONE gemm3 shape repeatedly (for rocprofv3 --pmc):  python tools/gemm_one.py M N K epi split cfg ver"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine
M, N, K, epi, split, cfg, ver = (int(a) for a in sys.argv[1:8])
eng = HipEngine(0, "bf16x3")
us = C.c_float()
assert eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, split, cfg, 30, ver, C.byref(us)) == 0
print("avg us", us.value)
