"""GPU: the RCCL path of smalltts_amd.parallel executed for real on ONE GPU.  A process group is forced at
world size 1 (SMTTS_DIST_FORCE=1) so that init_process_group("nccl", device_id=...), the device-side all_gather_into_tensor of
fp32 audio and of PCM16-as-bytes, barrier, all_reduce(MAX) and destroy all run over RCCL — both under the driver's launcher
(`python -m torch.distributed.run --nproc-per-node 1`) and from a plain process.  The 1 -> 8 curve itself needs an 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_rccl_collectives_execute_at_world_one(launcher):
    env = dict(os.environ, SMTTS_DIST_BACKEND="nccl", SMTTS_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    cmd = [sys.executable, HELPER] if launcher == "plain" else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", env["MASTER_PORT"], HELPER]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r["backend"] == "nccl" and r["device"].startswith("cuda")
    assert r["f32"] and r["pcm16"] and r["ragged"] and r["max"] == 1.25 and r["destroyed"]


@pytest.mark.parametrize("gather", ["f32", "pcm16"])
def test_bench_under_the_drivers_launcher_with_the_rccl_gather_in_the_timed_region(gather):
    """bench.py --gpus 1 under torch.distributed.run with backend nccl: every step ends in the RCCL all-gather (world 1)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--gather", gather,
           "--no-roofline", "--no-cpu-baseline", "--min-seconds", "0.3"]
    env = dict(os.environ, SMTTS_DIST_BACKEND="nccl", SMTTS_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 1 and r["dist"] == {"backend": "nccl", "collective": True, "world": 1}
    assert r["value"] > 100 and abs(r["value"] - 80.0 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3


def test_three_all_gathers_in_flight_on_three_streams_through_one_rccl_communicator():
    """batches in flight x collective.  200 rounds of kernel -> all_gather_into_tensor on three streams sharing
    the process group's single RCCL communicator; every slot's buffer is bit-checked before it is reused."""
    env = dict(os.environ, SMTTS_DIST_BACKEND="nccl", SMTTS_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_inflight.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r["backend"] == "nccl" and r["rounds"] == 200 and r["slots"] == 3
    assert r["mismatches"] == 0 and r["buffers_reused"]
