"""Worker for tests/test_parallel_cpu.py: launched by `python -m torch.distributed.run` exactly as the driver launches
bench.py at N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), it walks the same ShardContext calls
bench.py makes — from_env, barrier, steady-state gather into a pre-allocated buffer, max_over_ranks, close — with a
deterministic stand-in for the HIP engine (which needs a GPU), and prints one JSON line on rank 0."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smalltts_amd.parallel import ShardContext, synthesize_sharded  # noqa: E402

B, S = 4, 3200 * 3


def fake_audio(rank, step):
    g = torch.Generator().manual_seed(1000 * rank + step)
    return torch.randn(B, 1, S, generator=g)


def main():
    ctx = ShardContext.from_env()
    assert ctx.world == int(os.environ["WORLD_SIZE"]) and ctx.rank == int(os.environ["RANK"])
    pcm = len(sys.argv) > 1 and sys.argv[1] == "pcm16"
    buf = ctx.gather_buffer(ctx.world * B, S, torch.int16 if pcm else torch.float32)
    ctx.barrier()
    t0 = time.perf_counter()
    ok = True
    for step in range(3):
        local = fake_audio(ctx.rank, step)
        if pcm:
            local = (local.clamp(-1, 1) * 32767).round().to(torch.int16)
        full = ctx.gather_waveforms(local, ctx.world * B, out=buf)
        for r in range(ctx.world):
            want = fake_audio(r, step)
            if pcm:
                want = (want.clamp(-1, 1) * 32767).round().to(torch.int16)
            ok = ok and torch.equal(full[r * B:(r + 1) * B], want)
    ctx.barrier()
    dt = ctx.max_over_ranks(time.perf_counter() - t0 + 0.01 * ctx.rank)   # rank-dependent on purpose: MAX must win
    # ragged request list through the library entry point (what SmallTTS.synthesize_sharded calls)
    n = 2 * ctx.world + 1
    refs = [np.full((3, 64), i, np.float32) for i in range(n)]
    ids = [[i + 1] for i in range(n)]
    got = synthesize_sharded(lambda r, p, d: [np.full((1, 3200), float(x[0, 0]), np.float32) for x in r], refs, ids, 0.14, ctx=ctx)
    ok = ok and got.shape == (n, 1, 3200) and all(float(got[i, 0, 0]) == i for i in range(n))
    if ctx.rank == 0:
        print(json.dumps({"ok": bool(ok), "world": ctx.world, "backend": ctx.backend, "max_s": dt}))
    ctx.close()


if __name__ == "__main__":
    main()
