"""Host time to ENQUEUE one headline batch (every launch asynchronous) against the GPU time per batch, at the shipped precision:

    python tools/host_enqueue.py [--in-flight 1|3] [--procs 1|2] [--graph]

  --in-flight 3   bench.py's issue pattern (throughput tuning, batch i whole on stream i % 3, own workspace)
  --procs 2       two processes share the GPU (what two ranks of one host would do to the host side: each has its own
                  interpreter, HIP runtime and command queues) — prints one line per process
  --graph         also captures one batch (cond_encode -> sample -> codec_decode, latency tuning) into a HIP graph through
                  torch.cuda.CUDAGraph and replays it: host cost of one graph launch, and the batch time with the per-kernel
                  dispatch packets replaced by the graph's

SURVEY 8(e): ">= 6x at 8 GPUs requires host-side launch work not to serialise" — one process per GPU means each rank enqueues
only its own ~600 launches per batch; this tool says how much of a batch's GPU time one host thread needs for that."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-flight", type=int, default=1)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--batches", type=int, default=30)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.procs > 1:
        cmd = [sys.executable, os.path.abspath(__file__), "--in-flight", str(a.in_flight), "--precision", a.precision,
               "--batches", str(a.batches)]
        ps = [subprocess.Popen(cmd + ["--tag", f"proc {i} of {a.procs}: "], stdout=subprocess.PIPE, text=True) for i in range(a.procs)]
        for p in ps:
            print(p.communicate()[0].strip())
        return

    import torch
    import bench
    from smalltts_amd.engine import HipEngine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = HipEngine(0, a.precision)
    eng.load_synthetic(bench.SEED, parts=("dit", "decoder"))
    eng.finalize()
    inp = bench.make_inputs(dev, 0)
    k = max(1, a.in_flight)
    streams = [torch.cuda.Stream(dev) for _ in range(k)] if k > 1 else []

    def run(n, seed0):
        """Returns the host seconds spent inside one_step for each batch."""
        per = []
        if k <= 1:
            for i in range(n):
                t = time.perf_counter()
                bench.one_step(eng, inp, seed0 + i)
                per.append(time.perf_counter() - t)
            return per
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        prev = eng.set_tuning("throughput")
        try:
            for i in range(n):
                with torch.cuda.stream(streams[i % k]):
                    eng.use_workspace(f"batch{i % k}")
                    t = time.perf_counter()
                    bench.one_step(eng, inp, seed0 + i)
                    per.append(time.perf_counter() - t)
        finally:
            eng.use_workspace(None)
            eng.set_tuning(prev)
        for s in streams:
            cur.wait_stream(s)
        return per

    run(6, 0)
    torch.cuda.synchronize()
    n = a.batches
    t0 = time.perf_counter()
    per = run(n, 100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host, gpu = 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n
    # the queue is never empty while the host is ahead: "host" is the enqueue cost only when it is well below "gpu"; when the
    # two are equal the host is the bottleneck (or blocked on a full queue) — the per-batch minimum is the unblocked cost
    print(f"{a.tag}precision {a.precision}, {k} in flight: host enqueue {host:.2f} ms / batch (min {1e3 * min(per):.2f}, median "
          f"{1e3 * sorted(per)[len(per) // 2]:.2f}, max {1e3 * max(per):.2f}); all {n} batches done after {gpu:.2f} ms / batch; "
          f"host / GPU = {host / gpu:.2f}")

    if a.graph:
        # one batch as a HIP graph: static inputs, fixed sampler seed (a kernel argument, so baked into the graph)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            eng.use_workspace("graph")
            bench.one_step(eng, inp, 7)         # allocate the workspace outside the capture
            torch.cuda.synchronize()
            try:
                with torch.cuda.graph(g, stream=side):
                    out = bench.one_step(eng, inp, 7)
            except Exception as e:                  # noqa: BLE001
                print(f"{a.tag}graph capture failed: {type(e).__name__}: {e}")
                return
            finally:
                eng.use_workspace(None)
        torch.cuda.synchronize()
        ref = bench.one_step(eng, inp, 7).clone()
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hl = []
        for _ in range(n):
            t = time.perf_counter()
            g.replay()
            hl.append(time.perf_counter() - t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t3 = time.perf_counter()
        for i in range(n):
            bench.one_step(eng, inp, 7)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"{a.tag}HIP graph of one batch (latency tuning): replay {1e3 * (t2 - t0) / n:.2f} ms / batch, host {1e3 * (t1 - t0) / n:.3f} ms "
              f"per graph launch (min {1e3 * min(hl):.3f}); the same batch launched kernel by kernel {1e3 * (t4 - t3) / n:.2f} ms; "
              f"replayed output bit-identical to the eager one: {same}")


if __name__ == "__main__":
    main()
