#!/usr/bin/env python
"""GEMM microbenchmark on the hot-path shapes (run on the GPU box):  python tools/gemm_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine  # noqa: E402

SHAPES = [  # name, M, N, K, epi
    ("dit.qkvg", 600, 3840, 960, 0), ("dit.out", 600, 960, 960, 3), ("dit.ff1", 600, 4800, 960, 2),
    ("dit.ff2", 600, 960, 2432, 3), ("codec.s0.ffn1", 600, 8192, 2048, 1), ("codec.s3.ffn1", 120000, 1024, 256, 1),
    ("codec.s3.ffn2", 120000, 256, 1024, 3), ("codec.s4.ffn1", 480000, 512, 128, 1),
    ("codec.s5.ffn1", 960000, 256, 64, 1), ("codec.up1", 600, 8192, 4096, 0),
]


def main():
    eng = HipEngine(0, "bf16x3")
    cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["-1"])]
    print(f"{'shape':16s} {'M':>7s} {'N':>5s} {'K':>5s} ver cfg split {'us':>9s} {'TF(alg)':>8s} {'GB/s(alg)':>9s}")
    for name, M, N, K, epi in SHAPES:
        for ver in ((1, 3) if os.environ.get('GB_V1') else (3,)):
            for cfg in cfgs:
                if epi == 2 and cfg != -1:
                    continue
                for split in ((3, 1) if not os.environ.get('GB_S3') else (3,)):
                    us = C.c_float()
                    rc = eng.lib.smtts_bench_gemm(eng.h, M, N, K, epi, split, cfg, 20, ver, C.byref(us))
                    if rc:
                        print(name, "error", eng.lib.smtts_last_error(eng.h).decode())
                        continue
                    fl = 2.0 * M * N * K
                    by = M * K * 4 + N * K * (4 if split == 3 else 2) + M * (N // 2 if epi == 2 else N) * 4 * (2 if epi == 3 else 1)
                    print(f"{name:16s} {M:7d} {N:5d} {K:5d} v{ver} {cfg:3d} {split:5d} {us.value:9.1f} "
                          f"{fl / us.value / 1e6:8.1f} {by / us.value / 1e3:9.1f}")


if __name__ == "__main__":
    main()
