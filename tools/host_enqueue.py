"""Host time to enqueue one headline batch (all launches asynchronous) vs the GPU time per batch:  python tools/host_enqueue.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from smalltts_amd.engine import HipEngine
torch.cuda.set_device(0)
eng = HipEngine(0, "bf16x3")
eng.load_synthetic(bench.SEED, parts=("dit", "decoder")); eng.finalize()
inp = bench.make_inputs(torch.device("cuda", 0), 0)
for i in range(3):
    bench.one_step(eng, inp, i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
per = []
for i in range(n):
    a = time.perf_counter()
    bench.one_step(eng, inp, 100 + i)
    per.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms / batch (min {1e3 * min(per):.2f}, max {1e3 * max(per):.2f}); all {n} done after {1e3 * (t2 - t0) / n:.2f} ms / batch")
