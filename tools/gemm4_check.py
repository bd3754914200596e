"""gemm4 (256 x 256 x 64, phase-split schedule) on the GPU box:  python tools/gemm4_check.py [check] [bench]
check: cfg 7 against fp64 torch and against gemm3's 128 x 128 tile on the same fp16 / bf16 operands, ragged and aligned shapes.
bench: the codec's wide products at fp16, cfg -1 (gemm3's pick) / 1 (128 x 128) / 7 (gemm4); epi 4 = GELU hidden written as one fp16
array, 5 = LayerScale residual into fp32 x, 0 = fp32 store."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, "f16")
what = sys.argv[1:] or ["check", "bench"]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


if "check" in what:
    bad = 0
    for (M, N, K) in [(256, 256, 128), (256, 256, 192), (333, 200, 128), (333, 200, 960), (600, 960, 2432), (4800, 1024, 512),
                      (1000, 520, 256), (24000, 512, 2048)]:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ref = (A.double() @ W.double().t() + b.double())
        for split, tol in ((2, 2e-3), (1, 1.2e-2)):
            o7 = eng.test_gemm3(A, W, b, split=split, cfg=7).cpu()
            o1 = eng.test_gemm3(A, W, b, split=split, cfg=1).cpu()
            e7, e1, d = rel(o7, ref), rel(o1, ref), rel(o7, o1)
            reps = [eng.test_gemm3(A, W, b, split=split, cfg=7).cpu() for _ in range(4)]
            same = all(torch.equal(o7, r) for r in reps)
            ok = e7 < tol and d < 2e-6 and same
            bad += not ok
            print(f"{M:6d}x{N:5d}x{K:5d} split {split}: gemm4 vs fp64 {e7:.2e}  gemm3 vs fp64 {e1:.2e}  gemm4 vs gemm3 {d:.2e}  "
                  f"repeatable {same}  {'ok' if ok else 'FAIL'}", flush=True)
        og = eng.test_gemm3(A, W, b, act="gelu", split=2, cfg=7).cpu()
        eg = rel(og, torch.nn.functional.gelu(ref.float()))
        print(f"   gelu: {eg:.2e}", flush=True)
        bad += not (eg < 3e-3)
    print("CHECK", "PASSED" if not bad else f"FAILED ({bad})")

if "bench" in what:
    SH = [("s2.ff1", 24000, 2048, 512, 4), ("s2.ff2", 24000, 512, 2048, 5), ("s1.ff1", 4800, 4096, 1024, 4), ("s1.ff2", 4800, 1024, 4096, 5),
          ("s0.ff1", 600, 8192, 2048, 4), ("s0.ff2", 600, 2048, 8192, 5),
          ("up.s2", 4800, 2560, 2048, 0), ("up.s3", 24000, 1280, 1024, 0), ("up.s4", 120000, 512, 512, 0),
          ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
    for name, M, N, K, epi in SH:
        row = []
        for cfg in (-1, 1, 7):
            us = C.c_float()
            rc = eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, 2, cfg, 20, 3, C.byref(us))
            row.append(f"cfg {cfg:2d}: {us.value:8.1f} us {2.0 * M * N * K / us.value / 1e6:7.1f} TF/s" if not rc else "error " + eng.lib.smtts_last_error(eng.h).decode())
        print(f"{name:8s} {M:6d}x{N:5d}x{K:5d} epi {epi} | " + " | ".join(row), flush=True)
