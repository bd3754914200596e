"""Worker for tests/test_parallel_cpu.py::test_world_8_*: the protocol of `bench.py --gpus 8` (BASELINE.json configs[3]: 64 x 10 s
sharded over 8 ranks) at the REAL sizes — 8 ranks x 8 utterances x 240000 samples, one pre-allocated gather buffer per batch in
flight (3 slots x 64 x 240000 fp32 = 184 MB per rank, allocated ONCE), slot i % 3 reused by batch i — with a stand-in for the HIP
engine.  Prints one JSON line on rank 0."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smalltts_amd.parallel import ShardContext  # noqa: E402

B, S, SLOTS, STEPS = 8, 240000, 3, 6


def fake_audio(rank, step):
    # cheap, deterministic, different in every (rank, step, utterance, sample-block)
    base = torch.arange(S, dtype=torch.float32).mul_(1e-6)
    return torch.stack([base + (1000 * rank + 10 * step + u) for u in range(B)])[:, None, :]


def main():
    torch.set_num_threads(1)
    ctx = ShardContext.from_env()
    assert ctx.world == int(os.environ["WORLD_SIZE"]) == 8 and ctx.rank == int(os.environ["RANK"])
    lo, hi = ctx.my_shard(ctx.world * B)
    ok = (lo, hi) == (ctx.rank * B, ctx.rank * B + B)                     # contiguous 8-utterance shards
    gathers = [ctx.gather_buffer(ctx.world * B, S) for _ in range(SLOTS)]   # bench.py: one per batch in flight, allocated once
    ptrs = [g.data_ptr() for g in gathers]
    nbytes = sum(g.numel() * g.element_size() for g in gathers)
    ctx.barrier()
    t0 = time.perf_counter()
    for step in range(STEPS):
        out = ctx.gather_waveforms(fake_audio(ctx.rank, step), ctx.world * B, out=gathers[step % SLOTS])
        ok = ok and out.data_ptr() == ptrs[step % SLOTS]                  # straight into the pre-allocated slot: nothing re-allocated
        for r in (0, ctx.rank, ctx.world - 1):                             # rows in global utterance order
            ok = ok and torch.equal(out[r * B:(r + 1) * B], fake_audio(r, step))
    ctx.barrier()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    ok = ok and [g.data_ptr() for g in gathers] == ptrs
    oks = torch.tensor([1.0 if ok else 0.0])
    import torch.distributed as dist
    dist.all_reduce(oks, op=dist.ReduceOp.MIN)
    if ctx.rank == 0:
        print(json.dumps({"ok": bool(oks.item() == 1.0), "world": ctx.world, "backend": ctx.backend, "gather_bytes_per_rank": nbytes,
                          "steps": STEPS, "max_s": dt}))
    ctx.close()


if __name__ == "__main__":
    main()
