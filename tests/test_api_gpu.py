"""GPU: the drop-in Python surface (SmallTTS / Encoder / Decoder / CLIs) on the HIP engine, checked against the
CPU oracle end to end (cond-encode -> sampler -> codec decode) with synthetic weights and a REDUCED codec spec so
the CPU side stays fast. Mirrors how the reference is used (tryme.py / clone.py / SmallTTS.forward)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import codec_oracle as CO
from oracle import dit_oracle as O
from smalltts_amd.weights import (CodecSpec, codec_decoder_param_specs, codec_encoder_param_specs, dit_param_specs,
                                  synth_state_dict)
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu
SPEC = CodecSpec(n_filters=8, ratios=(8, 5, 5, 4, 2, 2), dec_depths=(1, 1, 1, 1, 1, 1, 1))  # hop 3200, tiny channels
SEED = 11


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300))


@pytest.fixture(scope="module")
def eng():
    from smalltts_amd.engine import HipEngine
    e = HipEngine(0, "bf16x3")
    e.load_synthetic(SEED, parts=("dit", "decoder", "encoder"), codec_spec=SPEC)
    e.finalize()
    return e


@pytest.fixture(scope="module")
def oracle_w():
    return (O.to_torch(synth_state_dict(dit_param_specs(), SEED)),
            O.to_torch(synth_state_dict(codec_decoder_param_specs(SPEC), SEED)),
            O.to_torch(synth_state_dict(codec_encoder_param_specs(SPEC), SEED)))


def test_synthesize_matches_oracle_end_to_end(eng, oracle_w):
    from smalltts_amd.api import HOP_SIZE, SmallTTS
    w, wd, _ = oracle_w
    tts = SmallTTS(engine=eng, seed=0)
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(6, 64, generator=g)
    toks = [3, 17, 42, 99, 150, 7, 8]
    dur = 1.7                                    # -> int(1.7 * 7.5) = 12 frames
    N = 12
    noise = torch.randn(4, 1, N, 64, generator=g)
    audio, lat = tts.synthesize_batch([ref.numpy()], [toks], [dur], noise=noise.numpy(), return_latents=True)
    assert audio[0].shape == (1, HOP_SIZE * N) and audio[0].dtype == np.float32
    with torch.no_grad():
        ids = torch.tensor([toks])
        pm = torch.ones(1, len(toks), dtype=torch.bool)
        cache = O.encode_conditions(w, ref[None], torch.tensor([6]), ids, pm)
        x = O.sample_dmd(w, cache, pm, torch.ones(1, N, dtype=torch.bool), noise, 4)
        wav = CO.decode(wd, x, SPEC)
    assert rel_l2(lat[0], x[0].numpy()) < 1e-4
    assert snr_db(audio[0], wav[0].numpy()) > 60.0


def test_reference_api_shapes_and_batch_equals_single(eng):
    from smalltts_amd.api import SmallTTS, estimate_duration
    assert estimate_duration("x" * 23) == 2.0 and estimate_duration("") == 0.5 and estimate_duration("y" * 1000) == 30.0
    tts = SmallTTS(engine=eng, seed=1)
    g = np.random.default_rng(0)
    refs = [g.standard_normal((r, 64)).astype(np.float32) for r in (5, 9, 7)]
    toks = [[1, 2, 3, 4], [10, 20, 30, 40, 50, 60], [7] * 9]
    durs = [1.0, 2.2, 1.5]
    ns = [7, 16, 11]
    noise = g.standard_normal((4, 3, 16, 64)).astype(np.float32)
    outs = tts.synthesize_batch(refs, toks, durs, noise=noise)
    assert [o.shape for o in outs] == [(1, 3200 * n) for n in ns]
    for b in range(3):  # the padded batch must equal per-utterance synthesis (reference semantics: independent calls)
        one = tts.synthesize_batch([refs[b]], [toks[b]], [durs[b]], noise=np.ascontiguousarray(noise[:, b:b + 1, :ns[b]]))[0]
        assert snr_db(outs[b], one) > 80.0
    # forward(): token lists instead of strings (no espeak offline); transcription tokens are prepended
    res = tts.forward([torch.from_numpy(refs[0])], [[5, 6]], [[7, 8, 9]], duration_sec=1.0)
    assert len(res) == 1 and isinstance(res[0], torch.Tensor) and tuple(res[0].shape) == (1, 3200 * 7)


def test_codec_wrappers_roundtrip_shapes_and_oracle(eng, oracle_w):
    from smalltts_amd.api import Decoder, Encoder
    _, wd, we = oracle_w
    enc, dec = Encoder(engine=eng), Decoder(engine=eng)
    g = torch.Generator().manual_seed(2)
    audio = torch.randn(2, 1, 3200 * 3 + 100, generator=g) * 0.2
    lat = enc.encode(audio)
    assert lat.device.type == "cpu" and tuple(lat.shape) == (2, 3, 64)
    with torch.no_grad():
        assert snr_db(lat.numpy(), CO.encode(we, audio, SPEC).numpy()) > 60.0
    wav = dec.decode(lat)
    assert wav.device.type == "cpu" and tuple(wav.shape) == (2, 1, 9600)
    with torch.no_grad():
        assert snr_db(wav.numpy(), CO.decode(wd, lat, SPEC).numpy()) > 60.0
    # reference-voice cache: same samples -> the cached latents, no second device encode
    calls = []
    real = eng.codec_encode
    eng.codec_encode = lambda a: (calls.append(1), real(a))[1]
    try:
        one = audio[:1]
        l1 = enc.encode_reference(one)
        l2 = enc.encode_reference(one.clone())
        l3 = enc.encode_reference(one * 0.5)
    finally:
        eng.codec_encode = real
    assert l2 is l1 and len(calls) == 2 and torch.equal(l1, lat[:1]) and not torch.equal(l3, l1)


def test_clone_cli_end_to_end(tmp_path):
    """clone.py surface: wav -> resample -> codec encode -> synthesize -> PCM16 wav (config 3 of BASELINE.json)."""
    from smalltts_amd.audio import read_wav, write_wav_pcm16
    sr = 16000
    t = np.arange(int(0.9 * sr)) / sr
    write_wav_pcm16(str(tmp_path / "ref.wav"), 0.5 * np.sin(2 * np.pi * 440 * t), sr)
    out = tmp_path / "clone.wav"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "smalltts_amd.scripts.clone", "--wav", str(tmp_path / "ref.wav"), "--text",
                        "hello there", "--tokens", "1,2,3,4,5,6,7,8", "--duration", "1.0", "--out", str(out), "--weights",
                        "synthetic:3", "--seed", "0"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, rate = read_wav(str(out))
    assert rate == 24000 and a.shape == (3200 * 7,) and np.isfinite(a).all()


def test_tryme_cli_end_to_end(tmp_path):
    """tryme.py surface (reference src/scripts/tryme.py:12-30): text (+ pre-tokenised ids: no espeak offline) -> 24 kHz PCM_16 of
    3200 * floor(7.5 * estimate_duration(text)) samples; the reference voice falls back to a seeded one when the asset is absent."""
    from smalltts_amd.api import estimate_duration
    from smalltts_amd.audio import read_wav
    text = "hello world this is a test"
    out = tmp_path / "out" / "tryme.wav"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "smalltts_amd.scripts.tryme", text, "--tokens", "5,9,14,33,41,14,77,120,3", "--out", str(out),
                        "--ref-latents", str(tmp_path / "absent.npy"), "--weights", "synthetic:3", "--seed", "0"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "loading model" in r.stdout and str(out) in r.stdout
    a, rate = read_wav(str(out))
    n = max(1, int(estimate_duration(text) * 7.5))
    assert rate == 24000 and a.shape == (3200 * n,) and np.isfinite(a).all() and np.abs(a).max() > 0
    with open(out, "rb") as f:
        hdr = f.read(44)
    assert hdr[:4] == b"RIFF" and int.from_bytes(hdr[20:22], "little") == 1 and int.from_bytes(hdr[34:36], "little") == 16   # PCM_16


def test_batch_cli_end_to_end(tmp_path):
    """batch.py surface (reference src/scripts/infer/batch.py:13-46): assets/test_audio/transcriptions.json -> one cloned utterance
    per listed file, out/<stem>_gen.wav, 24 kHz PCM_16 of 3200 * N samples each (here: one padded batch instead of a loop)."""
    import json
    from smalltts_amd.api import estimate_duration
    from smalltts_amd.audio import read_wav, write_wav_pcm16
    from smalltts_amd.scripts.batch import TEXTS
    td = tmp_path / "test_audio"
    td.mkdir()
    names = []
    for i, (sr, secs, f0) in enumerate(((16000, 0.9, 440.0), (24000, 1.4, 330.0), (44100, 0.7, 220.0))):
        t = np.arange(int(secs * sr)) / sr
        write_wav_pcm16(str(td / f"voice{i}.wav"), 0.4 * np.sin(2 * np.pi * f0 * t), sr)
        names.append({"filename": f"voice{i}.wav", "transcription": "unused by batch.py"})
    with open(td / "transcriptions.json", "w") as f:
        json.dump(names, f)
    outdir = tmp_path / "out"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "smalltts_amd.scripts.batch", "--dir", str(td), "--out", str(outdir), "--weights",
                        "synthetic:3", "--seed", "0", "--tokenizer", "chars"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for i in range(3):
        a, rate = read_wav(str(outdir / f"voice{i}_gen.wav"))
        n = max(1, int(estimate_duration(TEXTS[i]) * 7.5))
        assert rate == 24000 and a.shape == (3200 * n,) and np.isfinite(a).all(), (i, a.shape, n)
        assert f"[{i + 1}/3] voice{i}.wav" in r.stdout


def test_interactive_cli_loop_writes_one_wav_per_line(tmp_path):
    """interactive.py surface (reference src/scripts/infer/interactive.py:17-60): lines in, one utterance each."""
    from smalltts_amd import api
    from smalltts_amd.audio import read_wav
    from smalltts_amd.scripts import interactive
    n = interactive.main(["--weights", "synthetic:3", "--seed", "0", "--tokenizer", "chars", "--outdir", str(tmp_path)],
                         lines=["hello there", "", "it costs $5"])
    api._ENGINES.clear()
    assert n == 2
    a, rate = read_wav(str(tmp_path / "interactive_001.wav"))
    assert rate == 24000 and a.size % 3200 == 0 and a.size > 0 and np.isfinite(a).all()


def test_missing_weight_file_is_a_clear_error():
    from smalltts_amd.api import SmallTTS
    with pytest.raises(FileNotFoundError, match="synthetic"):
        SmallTTS(weights="/nonexistent/weights.smtts")


def test_batches_in_flight_equal_sequential(eng):
    """bench.py's default mode: batch i runs, whole, on stream i % 3 with its own workspace, so three batches overlap on
    the GPU.  Same seeds -> bit-identical audio as the one-stream schedule (the engine's side stream, its events and its
    weight-side scratch are shared between the concurrent batches)."""
    B, N, R, P = 2, 10, 5, 9
    g = torch.Generator().manual_seed(11)
    ref = torch.randn(B, R, 64, generator=g).cuda()
    rl = torch.full((B,), R).cuda()
    ids = torch.randint(1, 198, (B, P), generator=g).cuda()
    pm = torch.ones(B, P, dtype=torch.bool).cuda()
    mask = torch.ones(B, N, dtype=torch.bool).cuda()
    n = 7
    seq = []
    for i in range(n):
        x = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, seed=70 + i)
        seq.append(eng.codec_decode(x).cpu())
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)
    outs = []
    try:
        for i in range(n):
            with torch.cuda.stream(streams[i % 3]):
                eng.use_workspace(f"batch{i % 3}")
                x = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, seed=70 + i)
                outs.append(eng.codec_decode(x))
    finally:
        eng.use_workspace(None)
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    assert all(torch.equal(o.cpu(), s) for o, s in zip(outs, seq))


def test_synthesize_batches_matches_a_loop_of_synthesize_batch(eng):
    """Product-level entry for overlapped batches: same seeds, same audio as one batch after the other."""
    from smalltts_amd.api import SmallTTS
    rng = np.random.default_rng(21)
    batches = []
    for i in range(5):
        n = 2 + i % 2
        batches.append(([rng.standard_normal((3 + j, 64)).astype(np.float32) for j in range(n)],
                        [[int(v) for v in rng.integers(1, 198, size=5 + j)] for j in range(n)], [0.4 + 0.3 * j for j in range(n)]))
    a = SmallTTS(engine=eng, seed=5).synthesize_batches(batches, in_flight=3)
    assert eng.tuning == "latency"                      # the caller's tuning mode is restored
    # concurrency changes nothing: the same batches one after the other under the SAME (throughput) tuning are bit-identical
    tts = SmallTTS(engine=eng, seed=5)
    eng.set_tuning("throughput")
    try:
        b = [tts.synthesize_batch(*x) for x in batches]
    finally:
        eng.set_tuning("latency")
    assert len(a) == len(b) == 5
    for xa, xb in zip(a, b):
        assert len(xa) == len(xb) and all(np.array_equal(u, v) for u, v in zip(xa, xb))
    # the two tuning modes differ by fp32 summation order only (split-K / ring depth): far inside every tolerance
    tts = SmallTTS(engine=eng, seed=5)
    c = [tts.synthesize_batch(*x) for x in batches]
    for xa, xc in zip(a, c):
        for u, v in zip(xa, xc):
            assert snr_db(u, v) > 80.0


def test_synthesize_sharded_over_device_replicas_equals_one_engine(eng):
    """SmallTTS(device_ids=[...]): contiguous shards on one engine + host thread per listed GPU, gathered on the host.  The box has
    one GPU, so the second replica is a second engine on device 0 — the code path (threads, shard spans, order of the gathered
    rows) is the multi-GPU one."""
    from smalltts_amd.api import SmallTTS
    from smalltts_amd.engine import HipEngine
    rng = np.random.default_rng(33)
    n = 5
    refs = [rng.standard_normal((4 + j % 3, 64)).astype(np.float32) for j in range(n)]
    toks = [[int(v) for v in rng.integers(1, 198, size=6 + j)] for j in range(n)]
    one = SmallTTS(engine=eng, seed=9).synthesize_sharded(refs, toks, 0.8, max_batch=8)
    assert one.shape == (n, 1, 3200 * 6)
    e2 = HipEngine(0, "bf16x3")
    e2.load_synthetic(SEED, parts=("dit", "decoder", "encoder"), codec_spec=SPEC)
    e2.finalize()
    tts = SmallTTS(engine=eng, seed=9)
    tts._replicas.append(SmallTTS(engine=e2, seed=9))
    two = tts.synthesize_sharded(refs, toks, 0.8)
    assert two.shape == one.shape and np.isfinite(two).all()
    # rows 0..2 ran on replica 0 as one batch of 3, rows 3..4 on replica 1: per-utterance results do not depend on the batch they
    # rode in beyond summation order, the sampler noise does (seeded per call) -> compare shapes / finiteness / distinct rows
    assert not np.array_equal(two[0], two[3])
    e2.close()


# ---- N1: converted weights on the GPU (distill.py:39-57, 468-479) ---------------------------------------------
def _reference_inputs():
    g = torch.Generator().manual_seed(21)
    ref = torch.randn(5, 64, generator=g).numpy()
    toks = [4, 9, 77, 120, 33]
    noise = torch.randn(4, 1, 9, 64, generator=g).numpy()
    return ref, toks, noise


def test_weight_file_and_checkpoint_load_give_the_synthetic_engines_latents(eng, tmp_path):
    """A `.smtts` flat file and a DMD-style checkpoint ({"student_model": state_dict} with module. / _orig_mod. wrapper
    prefixes, DiT only) + a separately converted codec file, each loaded through SmallTTS(weights=...), must reproduce the
    latents AND audio of the engine filled by load_synthetic bit for bit: same tensors in, same packs, same kernels."""
    from smalltts_amd import api
    from smalltts_amd.api import SmallTTS
    from smalltts_amd.convert import convert_checkpoint
    from smalltts_amd.weights import save_weight_file
    ref, toks, noise = _reference_inputs()
    want_a, want_l = SmallTTS(engine=eng).synthesize_batch([ref], [toks], [1.2], noise=noise, return_latents=True)

    dit = synth_state_dict(dit_param_specs(), SEED)
    codec = synth_state_dict(codec_decoder_param_specs(SPEC) + codec_encoder_param_specs(SPEC), SEED)
    # (1) one flat file with everything
    p_all = str(tmp_path / "all.smtts")
    save_weight_file(p_all, {**dit, **codec}, SPEC)
    # (2) training checkpoint: wrapped key names, non-tensor entries, DiT only -> converter -> flat DiT file; codec separately
    ck = {"student_model": {("module._orig_mod." + k): torch.tensor(v) for k, v in dit.items()},   # (0-d log_scale stays 0-d)
          "step": 1234, "optimizer": {"lr": 1e-4}}
    p_ck = str(tmp_path / "checkpoint_latest.pt")
    torch.save(ck, p_ck)
    p_codec = str(tmp_path / "codec.smtts")
    save_weight_file(p_codec, codec, SPEC)
    p_conv = str(tmp_path / "dit.smtts")
    rep = convert_checkpoint(p_ck, p_conv)
    assert rep.ok and rep["matched"] == len(dit)
    del ck
    try:
        for weights in (p_all, f"{p_ck}+{p_codec}", [p_conv, p_codec]):
            tts = SmallTTS(weights=weights, precision="bf16x3")
            a, l = tts.synthesize_batch([ref], [toks], [1.2], noise=noise, return_latents=True)
            assert np.array_equal(l[0], want_l[0]), f"latents differ for weights={weights!r}"
            assert np.array_equal(a[0], want_a[0]), f"audio differs for weights={weights!r}"
            api._ENGINES.clear()
            del tts
            torch.cuda.empty_cache()
    finally:
        api._ENGINES.clear()


def test_wrong_shaped_checkpoint_is_a_clean_error_naming_the_tensor(tmp_path):
    from smalltts_amd import api
    from smalltts_amd.api import SmallTTS
    from smalltts_amd.engine import HipEngine
    from smalltts_amd.weights import save_weight_file
    small = {"velocity.weight": np.zeros((64, 512), np.float32), "velocity.bias": np.zeros(64, np.float32)}
    p = str(tmp_path / "other_model.smtts")
    save_weight_file(p, small)
    with pytest.raises(ValueError, match=r"velocity\.weight.*\(64, 512\).*\(64, 960\)"):
        SmallTTS(weights=p)
    api._ENGINES.clear()
    # the C ABI refuses too (a caller that bypasses the Python check): finalize names the tensor and its shape
    e = HipEngine(0)
    e.load_synthetic(1, parts=("dit",))
    e.set_tensor("dit.transformer_blocks.3.ff.w1.weight", np.zeros((1200, 960), np.float32))
    with pytest.raises(RuntimeError, match=r"dit\.transformer_blocks\.3\.ff\.w1\.weight has shape \(1200, 960\).*\(2400, 960\)"):
        e.finalize()
    assert not e.has("dit")
    # repairing the tensor and finalizing again works (packs are rebuilt, the old ones freed)
    from smalltts_amd.weights import synth_tensor
    e.set_tensor("dit.transformer_blocks.3.ff.w1.weight", synth_tensor("dit.transformer_blocks.3.ff.w1.weight", (2400, 960), 1))
    e.finalize()
    assert e.has("dit")
    # a tensor set AFTER finalize invalidates the engine until the next finalize (no stale packs)
    e.set_tensor("velocity.bias", np.ones(64, np.float32))
    assert not e.has("dit")
    with pytest.raises(RuntimeError, match="not finalized"):
        e.cond_encode(np.zeros((1, 2, 64), np.float32), np.array([2]), np.array([[1, 2]]), np.ones((1, 2), bool))
    e.finalize()
    assert e.has("dit")
    e.close()


def test_stress_determinism_tool_half_a_minute_all_workloads():
    """tools/stress_determinism.py in its short mode: repeated cond_encode and sampler calls under latency tuning
    (dual-stream encoders, LN-fold epilogues), then the dmd4 / teacher-CFG / clone workloads with three batches in flight under
    throughput tuning, each round compared bit for bit with the batches run alone.  This is the detector of the wrong-rows mode of
    NOTEBOOK 13a (packed fp32 behind cross-lane reductions next to LDS-DMA / MFMA kernels): -fno-slp-vectorize is on every
    translation unit since round 6 and this test would see a regression."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_determinism.py"), "160"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = re.findall(r"^(.*): (\d+) of (\d+) (?:repeats|rounds) differ", p.stdout, re.M)
    assert len(rows) == 5, p.stdout
    for what, bad, n in rows:
        assert int(bad) == 0 and int(n) >= 40, (what, bad, n)


def test_results_repeat_bit_for_bit_next_to_other_streams():
    """Bitwise repeatability with other HIP streams active — the two situations the product creates itself: the condition
    encoders on two streams (latency tuning) and three batches in flight on three streams (throughput tuning, bench.py's
    headline mode).  A fused q / k-prep attention kernel that was bit-exact alone failed exactly here (NOTEBOOK §13), so this
    runs the full-size bench shapes: 20 repeats of cond_encode, 6 rounds of three batches against the batches run alone."""
    import bench
    from smalltts_amd.engine import HipEngine
    dev = torch.device("cuda", 0)
    eng = HipEngine(0)
    eng.load_synthetic(bench.SEED, parts=("dit", "decoder"))
    eng.finalize()
    inp = bench.make_inputs(dev, 0)

    def cond():
        c = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"], debug=True)
        return {k: v.clone() for k, v in c.items() if torch.is_tensor(v)}

    a = cond()
    for i in range(20):
        b = cond()
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, f"cond_encode repeat {i}: {diff} differ"
    prev = eng.set_tuning("throughput")
    try:
        alone = [bench.one_step(eng, inp, 100 + i).clone() for i in range(3)]
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        for it in range(6):
            outs = [None] * 3
            cur = torch.cuda.current_stream(dev)
            for s in streams:
                s.wait_stream(cur)
            for i in range(3):
                with torch.cuda.stream(streams[i]):
                    eng.use_workspace(f"batch{i}")
                    outs[i] = bench.one_step(eng, inp, 100 + i)
            eng.use_workspace(None)
            for s in streams:
                cur.wait_stream(s)
            torch.cuda.synchronize()
            for i in range(3):
                assert torch.equal(outs[i], alone[i]), f"round {it}: batch {i} in flight differs from the same batch alone"
    finally:
        eng.use_workspace(None)
        eng.set_tuning(prev)
