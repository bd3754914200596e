"""How should batches in flight be laid onto HIP streams?  (GPU box)   python tools/schedule_probe.py [steps]
  whole:  batch i runs cond-encode -> sampler -> decode on stream i % 3           (bench.py / SmallTTS.synthesize_batches today)
  phase:  F front streams run cond-encode + sampler of batch i (alternating), ONE codec stream decodes batch i behind an event
Throughput tuning, B = 8 x 10 s, ms per batch over `steps` batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from smalltts_amd.engine import HipEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0)
eng = HipEngine(0); eng.load_synthetic(bench.SEED, parts=("dit", "decoder")); eng.finalize()
inp = bench.make_inputs(dev, 0)
eng.set_tuning("throughput")


def whole(n, k):
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    cur = torch.cuda.current_stream(dev)
    for s in streams: s.wait_stream(cur)
    out = None
    for i in range(n):
        with torch.cuda.stream(streams[i % k]):
            eng.use_workspace(f"w{i % k}")
            out = bench.one_step(eng, inp, 100 + i)
    eng.use_workspace(None)
    for s in streams: cur.wait_stream(s)
    return out


def phase(n, F, C=1):
    fs = [torch.cuda.Stream(dev) for _ in range(F)]
    cs = [torch.cuda.Stream(dev) for _ in range(C)]
    cur = torch.cuda.current_stream(dev)
    for s in fs + cs: s.wait_stream(cur)
    out, keep = None, []
    for i in range(n):
        f, c = fs[i % F], cs[i % C]
        with torch.cuda.stream(f):
            eng.use_workspace(f"f{i % F}")
            cache = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
            x = eng.sample(cache, inp["mask"], num_steps=4, seed=100 + i)
            ev = torch.cuda.Event(); ev.record(f)
        with torch.cuda.stream(c):
            c.wait_event(ev)
            eng.use_workspace(f"c{i % C}")
            out = eng.codec_decode(x)
            x.record_stream(c)
        keep.append((cache, x, out))
        if len(keep) > 8: keep.pop(0)
    eng.use_workspace(None)
    for s in fs + cs: cur.wait_stream(s)
    return out


def clock(fn, *a):
    fn(6, *a); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(steps, *a); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


for name, fn, a in (("whole x3", whole, (3,)), ("whole x2", whole, (2,)), ("phase F=1 C=1", phase, (1, 1)), ("phase F=2 C=1", phase, (2, 1)),
                    ("phase F=2 C=2", phase, (2, 2)), ("phase F=3 C=1", phase, (3, 1)), ("whole x3 again", whole, (3,))):
    print(f"{name:18s} {clock(fn, *a):7.3f} ms per batch"); sys.stdout.flush()
