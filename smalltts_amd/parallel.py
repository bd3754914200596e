"""Data-parallel synthesis across the GPUs of one node — the ONE implementation of the N > 1 path: `SmallTTS.synthesize_sharded`,
`bench.py --gpus N` and the tests all go through `ShardContext`.

Model: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU / in tests),
utterances sharded contiguously, weights replicated, and NO collective on the data path except ONE all-gather of the
finished waveform shards.  xGMI is point-to-point (7 links per GPU), so a single large gather per batch (7.68 MB per rank
at 8 x 10 s of fp32 audio, half of that as PCM16) is the whole communication budget.  The reference has no multi-GPU
inference path at all (src/server/src/main.rs:24 serialises requests behind one mutex).

For callers that do not want a launcher, `SmallTTS(device_ids=[...])` runs the same contiguous shards on several GPUs from
one process (one engine and one host thread per GPU, results gathered on the host): `run_shards_in_threads`.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; earlier ranks take the remainder (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def _wire(t: torch.Tensor) -> torch.Tensor:
    """int16 PCM travels as bytes: neither RCCL / NCCL nor gloo has a 16-bit integer type."""
    return t.view(torch.uint8) if t.dtype == torch.int16 else t


def all_gather_waveforms(local: torch.Tensor, n_total: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """local: (n_local, 1, S) on this rank (fp32 audio or int16 PCM) -> (n_total, 1, S) on every rank, rows in global
    utterance order.  Equal shards: one all_gather_into_tensor straight into `out` (pre-allocated by steady-state callers);
    ragged shards are padded to the largest and trimmed."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    mx = max(sizes)
    S = local.shape[-1]
    if mx == 0:
        return local.new_zeros((0, 1, S))
    if min(sizes) == mx:
        if out is None:
            out = local.new_empty((world * mx, 1, S))
        assert out.shape == (world * mx, 1, S) and out.dtype == local.dtype and out.device == local.device
        dist.all_gather_into_tensor(_wire(out), _wire(local.contiguous()), group=group)
        return out
    pad = local.new_zeros((mx, 1, S))
    pad[: local.shape[0]] = local
    buf = local.new_empty((world * mx, 1, S))
    dist.all_gather_into_tensor(_wire(buf), _wire(pad), group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], 0)


class ShardContext:
    """This process's place in the one-process-per-GPU job: rank / world / device, the barrier + max-over-ranks timing the
    benchmark contract asks for, and the waveform gather.  world == 1 needs no process group and every method degenerates."""

    def __init__(self, world: int = 1, rank: int = 0, local_rank: int = 0, backend: Optional[str] = None, group=None,
                 owns_group: bool = False, collective: Optional[bool] = None):
        self.world, self.rank, self.local_rank = int(world), int(rank), int(local_rank)
        self.backend, self.group, self._owns = backend, group, owns_group
        # collectives are issued when there is more than one rank — or when a process group was FORCED at world 1
        # (SMTTS_DIST_FORCE=1): the RCCL path (init with device_id, device-side all_gather_into_tensor, barrier, all_reduce,
        # destroy) can then be executed and tested on a single GPU, so an 8-GPU run does not meet it for the first time
        self.collective = (self.world > 1) if collective is None else bool(collective)
        self.on_gpu = torch.cuda.is_available()
        n_dev = torch.cuda.device_count() if self.on_gpu else 0
        # gloo runs (tests, several ranks sharing one GPU) wrap the local rank; an nccl job has one GPU per rank
        self.device_index = (self.local_rank % n_dev if backend != "nccl" else self.local_rank) if n_dev else -1
        self.device = torch.device("cuda", self.device_index) if n_dev else torch.device("cpu")
        # collectives run on the GPU under nccl / RCCL, on host tensors under gloo
        self.comm_device = self.device if backend == "nccl" else torch.device("cpu")

    @classmethod
    def from_env(cls, backend: Optional[str] = None) -> "ShardContext":
        """Under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): join the
        job.  Without WORLD_SIZE > 1: a single-process context.  SMTTS_DIST_BACKEND overrides the backend (gloo: CPU tests,
        or 2 ranks sharing a 1-GPU box)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        force = os.environ.get("SMTTS_DIST_FORCE", "") == "1"   # a process group even at world 1 (tests / the nccl smoke run)
        if world <= 1 and not force:
            return cls(1, 0, local)
        backend = backend or os.environ.get("SMTTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL needs it on this driver)
        owns = False
        if not dist.is_initialized():
            if world <= 1:   # forced single-rank group outside a launcher: supply the rendezvous ourselves
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                os.environ.setdefault("WORLD_SIZE", "1")
                os.environ.setdefault("RANK", "0")
            if backend == "nccl":
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
            owns = True
        ctx = cls(max(world, 1), rank, local, backend, None, owns, collective=True)
        if ctx.on_gpu:
            torch.cuda.set_device(ctx.device_index)
        return ctx

    @classmethod
    def current(cls) -> "ShardContext":
        """The already-initialised default process group (library use), or a single-process context."""
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_world_size(), dist.get_rank(), int(os.environ.get("LOCAL_RANK", dist.get_rank())),
                       dist.get_backend())
        return cls()

    # ---- timing contract: barrier + device sync on both sides, MAX over ranks --------------------------------------
    def barrier(self) -> None:
        if self.on_gpu:
            torch.cuda.synchronize()
        if self.collective:
            dist.barrier(group=self.group)
            if self.on_gpu:
                torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if not self.collective:
            return float(seconds)
        t = torch.tensor([seconds], device=self.comm_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    # ---- data path ----------------------------------------------------------------------------------------------------
    def my_shard(self, n_items: int) -> Tuple[int, int]:
        return shard_range(n_items, self.world, self.rank)

    def gather_buffer(self, n_total: int, samples: int, dtype=torch.float32) -> Optional[torch.Tensor]:
        """Pre-allocated output of gather_waveforms for a steady-state loop with equal shards (None without collectives)."""
        if not self.collective:
            return None
        return torch.empty(n_total, 1, samples, dtype=dtype, device=self.comm_device)

    def gather_waveforms(self, local: torch.Tensor, n_total: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(n_local, 1, S) of this rank -> (n_total, 1, S) on every rank; the only collective of the path."""
        if not self.collective:
            return local
        if local.device != self.comm_device:
            local = local.to(self.comm_device)
        return all_gather_waveforms(local, n_total, self.group, out)

    def close(self) -> None:
        if self.collective and self._owns and dist.is_initialized():
            dist.barrier(group=self.group)
            dist.destroy_process_group()


def synthesize_sharded(synth_fn: Callable, ref_latents: Sequence[np.ndarray], phoneme_ids: Sequence[Sequence[int]],
                       duration_sec: float, device=None, group=None, ctx: Optional[ShardContext] = None) -> torch.Tensor:
    """Every rank holds the full request list, synthesises its contiguous shard with
    `synth_fn(refs, ids, duration) -> list of (1, S) arrays or a (n, 1, S) tensor` (e.g. SmallTTS.synthesize_batch) and
    returns the complete (n, 1, S) waveform batch after one all-gather.  All utterances share `duration_sec` so S is
    uniform (the batch = 64 x 10 s configuration of BASELINE.json)."""
    ctx = ctx or ShardContext.current()
    if group is not None:
        ctx.group = group
    n = len(ref_latents)
    lo, hi = ctx.my_shard(n)
    S = max(1, int(duration_sec * 24_000 / 3_200)) * 3_200
    dev = torch.device(device) if device is not None else ctx.comm_device
    if hi > lo:
        outs = synth_fn(list(ref_latents[lo:hi]), list(phoneme_ids[lo:hi]), duration_sec)
        if isinstance(outs, torch.Tensor):
            local = outs.to(dev)
        else:
            local = torch.from_numpy(np.stack([np.asarray(o, np.float32) for o in outs])).to(dev)
    else:
        local = torch.zeros((0, 1, S), device=dev)
    if ctx.world <= 1:
        return local
    return all_gather_waveforms(local, n, ctx.group)


def run_shards_in_threads(shard_fns: Sequence[Callable[[int, int], list]], n_items: int) -> list:
    """In-process variant: replica r (one engine on its own GPU) runs shard_range(n_items, len(shard_fns), r) on its own host
    thread — ctypes releases the GIL inside the library and each engine enqueues on its own device — and the per-utterance
    results come back in request order.  No collective: the shards meet on the host."""
    world = len(shard_fns)
    spans = [shard_range(n_items, world, r) for r in range(world)]
    with ThreadPoolExecutor(max_workers=world) as pool:
        futs = [pool.submit(fn, lo, hi) if hi > lo else None for fn, (lo, hi) in zip(shard_fns, spans)]
        outs: list = []
        for f in futs:
            if f is not None:
                outs.extend(f.result())
    return outs
