"""Per-phase / per-kernel time of one headline batch (HIP events, tagged profiler mode).

    python tools/phase_breakdown.py [--workload dmd4|clone] [--reps 5]
Prints ms per batch grouped by pipeline phase (enc, mod, dit, dec.s<i>, cenc.s<i>) and the kernels inside each.
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dmd4")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    from smalltts_amd.engine import HipEngine
    torch.cuda.set_device(0)
    eng = HipEngine(0, a.precision)
    eng.load_synthetic(bench.SEED, parts=("dit", "decoder", "encoder") if a.workload == "clone" else ("dit", "decoder"))
    eng.finalize()
    inp = bench.make_inputs(torch.device("cuda", 0), 0)
    for i in range(2):
        bench.one_step(eng, inp, 10 + i, workload=a.workload)
    torch.cuda.synchronize()
    eng.profile(True, tagged=True)
    for i in range(a.reps):
        bench.one_step(eng, inp, 900 + i, workload=a.workload)
    torch.cuda.synchronize()
    rows = eng.profile_report()
    eng.profile(False)
    phases = collections.defaultdict(list)
    for r in rows:
        ph, _, k = r["name"].partition("/") if "/" in r["name"] else ("-", "", r["name"])
        phases[ph].append((r["ms"] / a.reps, r["launches"] // a.reps, k, r["flops"] / a.reps, r["bytes"] / a.reps))
    tot = sum(x[0] for v in phases.values() for x in v)
    print(f"total kernel time {tot:.3f} ms / batch")
    for ph, v in sorted(phases.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        pm = sum(x[0] for x in v)
        pf = sum(x[3] for x in v)
        print(f"\n[{ph}] {pm:.3f} ms ({100 * pm / tot:.1f} %)  {pf / 1e9:.1f} GFLOP  -> {pf / pm / 1e9:.1f} TFLOP/s algorithmic")
        for ms, n, k, fl, by in sorted(v, key=lambda x: -x[0]):
            print(f"   {ms:8.3f} ms  x{n:<4d} {1e3 * ms / max(n, 1):8.1f} us  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s {by / max(ms, 1e-9) / 1e6:8.1f} GB/s  {k}")


if __name__ == "__main__":
    main()
