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


def _bench(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in drop:
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=300,
                          cwd=root, env=env)


def test_bench_gpus_n_without_enough_gpus_is_a_hard_error_not_a_world_1_line():
    """`python bench.py --gpus 8` outside a launcher used to benchmark ONE GPU and print n_gpus: 1.  Without N visible
    GPUs (this container has none) it must exit non-zero and print no JSON line at all."""
    p = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 8" in p.stderr and "GPUs" in p.stderr


def test_bench_gpus_n_disagreeing_with_the_launchers_world_is_a_hard_error():
    p = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0",
                                                                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}, drop=())
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "WORLD_SIZE=2" in p.stderr and "--nproc-per-node 4" in p.stderr


def test_bench_relaunch_command_is_the_drivers_launcher_line(monkeypatch):
    """ensure_world() with enough devices execs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py <same argv>` (the exec itself is stubbed out)."""
    import bench
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench.os, "execvpe", lambda f, a, e: seen.update(file=f, argv=a, env=e))
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    bench.ensure_world(8, ["--gpus", "8", "--steps", "20", "--warmup", "3"])
    a = seen["argv"]
    assert a[1:6] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8"] and "--master-addr" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    assert a[-7].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen.clear()
    bench.ensure_world(1, ["--gpus", "1"])          # N = 1 outside a launcher: nothing to do
    assert not seen


def test_world_8_ranks_walk_the_bench_protocol_at_the_64_utterance_size():
    """BASELINE.json configs[3] (64 x 10 s over 8 GPUs) without 8 GPUs: the driver's launcher line at --nproc-per-node 8 over
    gloo, bench.py's ShardContext calls at the real sizes (8-utterance contiguous shards, 3 gather slots of 64 x 240000 fp32 =
    184 MB per rank allocated once and reused, rows in global utterance order, MAX-over-ranks timing)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "helpers", "shard_worker8.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["world"] == 8 and res["backend"] == "gloo"
    assert res["gather_bytes_per_rank"] == 3 * 64 * 240000 * 4
