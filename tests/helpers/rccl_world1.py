"""Run by tests/test_parallel_gpu.py: everything ShardContext does over the nccl (= RCCL) backend, at world size 1 on one GPU —
init_process_group("nccl", device_id=...), the device-side all_gather_into_tensor of fp32 audio and of the PCM16 uint8 view,
barrier, max_over_ranks (all_reduce MAX), close — so the 8-GPU job does not meet any of it for the first time."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smalltts_amd.parallel import ShardContext

os.environ.setdefault("SMTTS_DIST_FORCE", "1")
backend = os.environ.get("SMTTS_DIST_BACKEND", "nccl")
ctx = ShardContext.from_env()
assert ctx.collective and ctx.world == 1 and dist.is_initialized() and dist.get_backend() == backend, (ctx.world, ctx.collective)
dev = ctx.comm_device
g = torch.Generator().manual_seed(3)
audio = torch.randn(8, 1, 24000, generator=g).to(dev)
out = ctx.gather_buffer(8, 24000)
assert out is not None and out.device == dev
res = ctx.gather_waveforms(audio, 8, out=out)
ctx.barrier()
ok_f32 = bool(torch.equal(res, audio)) and res.data_ptr() == out.data_ptr()
pcm = (audio.clamp(-1, 1) * 32767).to(torch.int16)
out16 = ctx.gather_buffer(8, 24000, torch.int16)
res16 = ctx.gather_waveforms(pcm, 8, out=out16)
ctx.barrier()
ok_pcm = bool(torch.equal(res16, pcm)) and res16.dtype == torch.int16
ragged = ctx.gather_waveforms(audio[:5], 5)          # no pre-allocated buffer
ok_rag = bool(torch.equal(ragged, audio[:5]))
t = ctx.max_over_ranks(1.25)
ctx.close()
print(json.dumps({"backend": backend, "device": str(dev), "f32": ok_f32, "pcm16": ok_pcm, "ragged": ok_rag, "max": t,
                  "destroyed": not dist.is_initialized()}))
