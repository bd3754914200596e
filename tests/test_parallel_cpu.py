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


@pytest.mark.parametrize("mode", ["f32", "pcm16"])
def test_torchrun_two_ranks_walk_the_bench_protocol(mode):
    """The driver's N > 1 command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P <script>) with world 2 over gloo: rendezvous from the environment, barriers, the steady-state gather into a
    pre-allocated buffer (fp32 and int16 PCM), MAX-over-ranks timing and the ragged library path — ShardContext is the single
    implementation bench.py and SmallTTS.synthesize_sharded use."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "helpers", "shard_worker.py"), mode]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["world"] == 2 and res["backend"] == "gloo"
    assert res["max_s"] >= 0.01   # rank 1 reported 10 ms more than rank 0: the MAX over ranks was taken


def test_forced_world_one_group_runs_every_collective_over_gloo():
    """SMTTS_DIST_FORCE=1: the helper the GPU suite runs over nccl / RCCL (tests/helpers/rccl_world1.py), here over gloo on the
    CPU — a process group at world size 1, gather into a pre-allocated buffer (fp32 and PCM16 as bytes), barrier, all_reduce."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMTTS_DIST_BACKEND="gloo", SMTTS_DIST_FORCE="1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "rccl_world1.py")], capture_output=True, text=True,
                       timeout=300, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "gloo" and r["f32"] and r["pcm16"] and r["ragged"] and r["max"] == 1.25 and r["destroyed"]
