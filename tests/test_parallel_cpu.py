"""CPU, world_size 2 over gloo: the N>1 path (utterance sharding + waveform all-gather) reassembles
exactly what a single process produces.  The synthesis function is a deterministic stand-in (the HIP
engine needs a GPU); what is under test is the sharding/collective logic bench.py --gpus N relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smalltts_amd.parallel import all_gather_waveforms, shard_range, shard_sizes, synthesize_sharded


def _fake_synth(refs, ids, dur):
    S = int(dur * 7.5) * 3200
    outs = []
    for r, t in zip(refs, ids):
        seed = int(abs(float(np.sum(r))) * 1000) % 100000 + sum(t)
        outs.append(np.random.default_rng(seed).standard_normal((1, S)).astype(np.float32))
    return outs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        refs = [rng.standard_normal((5 + i % 3, 64)).astype(np.float32) for i in range(n)]
        ids = [[1 + (i * 7 + j) % 190 for j in range(4 + i % 5)] for i in range(n)]
        full = synthesize_sharded(_fake_synth, refs, ids, 0.4, torch.device("cpu"))
        q.put((rank, full.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1


@pytest.mark.parametrize("n", [8, 5])  # even shards (one collective) and ragged shards (padded)
def test_two_ranks_reassemble_single_process_result(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    refs = [rng.standard_normal((5 + i % 3, 64)).astype(np.float32) for i in range(n)]
    ids = [[1 + (i * 7 + j) % 190 for j in range(4 + i % 5)] for i in range(n)]
    want = np.stack(_fake_synth(refs, ids, 0.4))
    for r in range(2):
        assert got[r].shape == want.shape and np.array_equal(got[r], want)
