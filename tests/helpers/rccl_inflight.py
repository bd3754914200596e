"""Run by tests/test_parallel_gpu.py: what an 8-GPU `bench.py` run meets first — THREE batches in flight on three HIP streams, each
ending in an all-gather through the ONE RCCL communicator of the process group (world 1 here, SMTTS_DIST_FORCE=1), 200 rounds.
Each slot has its own pre-allocated gather buffer; a slot's buffer is checked (bit-equality with what that round put in) before the
slot is reused, so a collective that overtook its stream's producer kernel, or landed in another slot's buffer, is seen."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smalltts_amd.parallel import ShardContext

os.environ.setdefault("SMTTS_DIST_FORCE", "1")
ctx = ShardContext.from_env()
assert ctx.collective and ctx.world == 1 and dist.get_backend() == os.environ.get("SMTTS_DIST_BACKEND", "nccl")
dev, B, S, SLOTS, ROUNDS = ctx.comm_device, 8, 240000, 3, 200
streams = [torch.cuda.Stream(dev) for _ in range(SLOTS)]
gathers = [ctx.gather_buffer(B, S) for _ in range(SLOTS)]
ptrs = [g.data_ptr() for g in gathers]
base = torch.arange(B * S, device=dev, dtype=torch.float32).view(B, 1, S)
pending = [None] * SLOTS          # (round, event) of the gather last issued on the slot
bad = 0
cur = torch.cuda.current_stream(dev)
for s in streams:
    s.wait_stream(cur)
for i in range(ROUNDS):
    k = i % SLOTS
    with torch.cuda.stream(streams[k]):
        if pending[k] is not None:                       # same stream: ordered behind the previous gather into this slot
            j = pending[k]
            bad += int(not torch.equal(gathers[k], base * 0.5 + float(j)))
        audio = base * 0.5 + float(i)                    # the "decode" of round i: a kernel on this slot's stream
        out = ctx.gather_waveforms(audio, B, out=gathers[k])
        assert out.data_ptr() == ptrs[k]
        pending[k] = i
for k, s in enumerate(streams):
    cur.wait_stream(s)
torch.cuda.synchronize()
for k in range(SLOTS):
    bad += int(not torch.equal(gathers[k], base * 0.5 + float(pending[k])))
ctx.barrier()
ctx.close()
print(json.dumps({"backend": dist.get_backend() if dist.is_initialized() else os.environ.get("SMTTS_DIST_BACKEND", "nccl"),
                  "rounds": ROUNDS, "slots": SLOTS, "mismatches": bad, "buffers_reused": [g.data_ptr() for g in gathers] == ptrs}))
