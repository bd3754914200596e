"""What each phase costs when THREE batches of it are in flight (round-robin on three streams), next to one at a time:
    python tools/inflight_phases.py
Tells which phase bounds the in-flight throughput of the whole path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from smalltts_amd.engine import HipEngine
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = HipEngine(0, "bf16x3")
eng.load_synthetic(bench.SEED, parts=("dit", "decoder")); eng.finalize()
inp = bench.make_inputs(dev, 0)
cache = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
x = eng.sample(cache, inp["mask"], num_steps=bench.DMD_STEPS, seed=1)
phases = {
    "cond_encode": lambda i: eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"]),
    "sample(4 steps)": lambda i: eng.sample(cache, inp["mask"], num_steps=bench.DMD_STEPS, seed=i),
    "codec_decode": lambda i: eng.codec_decode(x),
}
streams = [torch.cuda.Stream(dev) for _ in range(3)]
def run(fn, n, nfl):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if nfl == 1:
        for i in range(n): fn(i)
    else:
        eng.set_dual_stream(False)
        for i in range(n):
            with torch.cuda.stream(streams[i % nfl]):
                eng.use_workspace(f"batch{i % nfl}")
                fn(i)
        eng.use_workspace(None); eng.set_dual_stream(True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
tot1 = tot3 = 0.0
for name, fn in phases.items():
    run(fn, 3, 1); run(fn, 6, 3)
    a, b = run(fn, 12, 1), run(fn, 24, 3)
    tot1 += a; tot3 += b
    print(f"{name:18s} one at a time {a:7.3f} ms   three in flight {b:7.3f} ms per batch")
print(f"{'sum':18s} one at a time {tot1:7.3f} ms   three in flight {tot3:7.3f} ms per batch")
