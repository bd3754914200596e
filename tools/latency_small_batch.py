"""Latency of the whole hot path (condition encoders -> 4-step DMD sampler -> codec decode) at small batch sizes, 10-s utterances:
    python tools/latency_small_batch.py [--batches 1 2 4 8] [--reps 30]
One batch at a time on one stream (latency tuning), HIP-event free: wall clock around `reps` calls with a device sync."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    from smalltts_amd.engine import HipEngine
    torch.cuda.set_device(0)
    eng = HipEngine(0, a.precision)
    eng.load_synthetic(bench.SEED, parts=("dit", "decoder"))
    eng.finalize()
    full = bench.make_inputs(torch.device("cuda", 0), 0)
    for nb in a.batches:
        inp = {k: (v[:nb].contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == bench.B else v) for k, v in full.items()}
        for i in range(3):
            bench.one_step(eng, inp, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.reps):
            bench.one_step(eng, inp, 100 + i)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.reps
        print(f"batch {nb} x 10 s: {ms:7.3f} ms  -> {nb * bench.AUDIO_SEC_PER_UTT / ms * 1e3:8.1f} audio-s/s, RTF {ms / 1e3 / (nb * bench.AUDIO_SEC_PER_UTT):.2e}")


if __name__ == "__main__":
    main()
