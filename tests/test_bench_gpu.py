"""GPU: bench.py itself — the JSON contract of the driver's command lines, at N = 1 and at N = 2 (two ranks share the box's
single GPU over gloo, SMTTS_DIST_BACKEND=gloo: same ShardContext code path as the 8-GPU RCCL job, host-side collective)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "value_sequential"}


def _last_json(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_single_gpu_line_has_the_contract_fields_and_rooflines():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _last_json(p.stdout)
    assert KEYS <= set(r) and r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["unit"] == "audio-seconds/sec" and r["value"] > 100 and r["value_sequential"] > 100 and r["vs_baseline"] is None
    assert abs(r["value"] - 80.0 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3        # 8 x 10 s per step
    assert r["config"]["workload"].startswith("cond-encode + 4-step DMD sampler + codec decode") and "model" not in r["config"]
    # the K-step region is repeated until >= 2 s have been timed; value / ms_per_step are the median repeat
    assert r["timed_seconds"] >= 2.0 and r["repeats"] >= 2 and r["ms_per_step_min"] <= r["ms_per_step"] <= r["ms_per_step_max"]
    assert r["dist"] == {"backend": None, "collective": False, "world": 1}
    # the roofline block describes the TIMED configuration: three batches in flight = throughput tuning; the latency-tuned twin
    # sits next to value_sequential
    assert r["config"]["tuning"] == r["roofline_tuning"] == "throughput"
    for rf, tuning in ((r["roofline"], "throughput"), (r["roofline_sequential"], "latency")):
        assert rf["tuning"] == tuning
        assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
        assert 0 < rf["frac_events"] < 1 and rf["avg_launch_us_events"] > 0 and rf["frac_source"]
        if rf["rocprof_avg_us"]:      # a hash-matching rocprofv3 summary is committed: frac follows from it
            assert abs(rf["avg_launch_us"] - rf["rocprof_avg_us"]) < 1e-3 and "rocprofv3" in rf["frac_source"]
        else:
            assert rf["frac"] == rf["frac_events"]
        assert rf["unit"] in ("GB/s", "TFLOP/s") and "traffic" in rf and len(rf["kernel_src_sha"]) == 16
    by_shape = r["roofline"].get("by_shape", [])
    assert all(t["avg_us_events"] > 0 and 0 < t["mfma_frac"] < 1 and "x" in t["MxNxK"] for t in by_shape) and len(by_shape) <= 3
    # the headline never launches the split-K partial-product class for the N = 960 projections (throughput tuning runs them unsplit)
    names = {k["name"] for k in r["kernel_breakdown"]}
    assert "splitk_resid_ln" not in names
    for ph in ("dit_sampler", "cond_encoders", "codec_decode"):
        assert r["phase_roofline"][ph]["ms_per_step"] > 0 and 0 < r["phase_roofline"][ph]["8d"]["mfma_frac"] < 1
        assert r["phase_roofline_sequential"][ph]["ms_per_step"] > 0
    # SURVEY 8(d) bytes of the codec: stage-boundary images once each way + weights once, derived from the spec
    assert abs(r["phase_roofline"]["codec_decode"]["8d"]["bytes"] - 2.57e9) < 0.02e9
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["cores_available"] >= cb["cores"] and cb["sample"]
    assert str(cb["cores"]) in cb["thread_probe_ms"]          # the thread count was chosen by the bounded probe
    # the other single-GPU configurations are clocked in the same run (BASELINE configs[2], configs[4], SURVEY 8(d) secondary point)
    sec = r["secondary"]
    assert set(sec) == {"clone", "teacher128", "dmd4_R38_P128"}
    for name, leg in sec.items():
        assert leg["ms_per_step"] > 0 and leg["sequential_ms_per_step"] > 0 and leg["unit"] == "audio-seconds/sec"
        assert abs(leg["value"] - 80.0 / (leg["ms_per_step"] * 1e-3)) / leg["value"] < 1e-3
        assert abs(leg["value_sequential"] - 80.0 / (leg["sequential_ms_per_step"] * 1e-3)) / leg["value_sequential"] < 1e-3
    assert sec["clone"]["ms_per_step"] > r["ms_per_step_min"] * 0.98          # + the codec encode of 8 x 2 s
    assert sec["teacher128"]["sequential_ms_per_step"] > 10 * r["sequential_ms_per_step"]   # 128 steps x 3B rows against 4 x B
    assert "R=38, P=128" in sec["dmd4_R38_P128"]["workload"]


@pytest.mark.parametrize("gather", ["f32", "pcm16"])
def test_bench_two_ranks_over_gloo_on_one_gpu(gather):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--gather", gather]
    env = dict(os.environ, SMTTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert KEYS <= set(r) and r["n_gpus"] == 2 and r["config"]["global_batch"] == 16 and r["scaling"] == "weak"
    assert abs(r["value"] - 160.0 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3       # both ranks' 8 x 10 s count
    assert "gloo" in r["config"]["parallelism"] and gather in r["config"]["parallelism"]
    assert "cpu_baseline" not in r                                                       # rank 0 at N = 1 only


def test_bench_eight_ranks_share_the_gpu_over_gloo_with_the_real_engine():
    """Config 4 first light (BASELINE configs[3]: 64 utterances over 8 ranks) as far as a 1-GPU box allows: `bench.py --gpus 8`
    outside a launcher re-executes under the driver's launcher line, eight ranks with the REAL engine share the box's GPU
    (8 x ~6.4 GB of weights + workspaces), the collective is gloo.  The line must say n_gpus 8 / global_batch 64, and the gathered
    waveform must hold the ranks' shards in global order: every rank draws its sampler noise from Philox(seed = step seed), its
    inputs from seed 1000 + rank, so row block r of the gathered batch equals what a world-1 run with rank r's inputs produces."""
    env = dict(os.environ, SMTTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SMTTS_BENCH_DUMP_ROWS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--min-seconds", "0", "--no-sequential",
              "--in-flight", "1", "--no-secondary"]
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + common, capture_output=True, text=True, timeout=2400, cwd=ROOT,
                       env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "re-executing" in p.stderr
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 8 and r["dist"] == {"backend": "gloo", "collective": True, "world": 8}
    assert r["config"]["global_batch"] == 64 and r["scaling"] == "weak" and "dp8" in r["config"]["parallelism"]
    assert abs(r["value"] - 640.0 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-3       # all eight ranks' 8 x 10 s count
    rows8 = r["row_checksums"]                                # one checksum per utterance of the LAST step's gathered batch
    assert len(rows8) == 64
    # world-1 runs standing in for rank 0, 3 and 7: same per-rank input seed, same step seeds -> the same 8 rows, bit for bit
    for rank in (0, 3, 7):
        env1 = dict(env, SMTTS_BENCH_RANK_SEED=str(rank))
        p1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, capture_output=True, text=True, timeout=900,
                            cwd=ROOT, env=env1)
        assert p1.returncode == 0, p1.stderr[-3000:]
        r1 = _last_json(p1.stdout)
        assert r1["n_gpus"] == 1 and len(r1["row_checksums"]) == 8
        assert r1["row_checksums"] == rows8[8 * rank: 8 * rank + 8], rank


def test_bench_gpus_2_without_a_launcher_relaunches_itself_as_two_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it must not print a world-1 line: it re-executes under the
    driver's launcher command (here two gloo ranks sharing the box's GPU)."""
    env = dict(os.environ, SMTTS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline",
                        "--no-cpu-baseline", "--min-seconds", "0"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "re-executing" in p.stderr
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 2 and r["dist"]["world"] == 2 and r["config"]["global_batch"] == 16
