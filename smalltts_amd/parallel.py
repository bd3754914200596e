"""Data-parallel synthesis across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in CPU tests), utterances sharded contiguously, weights
replicated, no collective on the data path except ONE all-gather of the waveform shards
(the reference has no multi-GPU inference path — src/server/src/main.rs:24 serialises
requests behind a mutex).  xGMI is point-to-point (7 links/GPU), so a single large all-gather per
batch (7.68 MB/rank at 8 x 10 s fp32) is the whole communication budget.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

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


def all_gather_waveforms(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local: (n_local, 1, S) on this rank -> (n_total, 1, S) on every rank, rows in global utterance
    order.  Equal shards use one all_gather_into_tensor; ragged shards are padded to the largest."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    mx = max(sizes)
    S = local.shape[-1]
    if mx == 0:
        return local.new_zeros((0, 1, S))
    if min(sizes) == mx:
        out = local.new_empty((world * mx, 1, S))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = local.new_zeros((mx, 1, S))
    pad[: local.shape[0]] = local
    buf = local.new_empty((world * mx, 1, S))
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], 0)


def synthesize_sharded(synth_fn, ref_latents: Sequence[np.ndarray], phoneme_ids: Sequence[Sequence[int]],
                       duration_sec: float, device, group=None) -> torch.Tensor:
    """Every rank holds the full request list, synthesises its contiguous shard with
    `synth_fn(refs, ids, duration) -> list of (1, S) arrays` (e.g. SmallTTS.synthesize_batch) and
    returns the complete (n, 1, S) waveform batch after one all-gather.  All utterances share
    `duration_sec` so S is uniform (the batch=64 x 10 s configuration of BASELINE.json)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(ref_latents)
    lo, hi = shard_range(n, world, rank)
    S = max(1, int(duration_sec * 24_000 / 3_200)) * 3_200
    if hi > lo:
        outs = synth_fn(list(ref_latents[lo:hi]), list(phoneme_ids[lo:hi]), duration_sec)
        local = torch.from_numpy(np.stack([np.asarray(o, np.float32) for o in outs])).to(device)
    else:
        local = torch.zeros((0, 1, S), device=device)
    return all_gather_waveforms(local, n, group)
