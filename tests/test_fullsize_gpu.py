"""GPU: parity at BASELINE.json's full sizes and at the edges of the reference's ranges
(estimate_duration caps at 30 s = 225 frames, infer/onnx.py:17-18; dummy data uses up to 198 phonemes / 64 reference
frames, data/dummy.py:16-32).  CPU oracle legs are sized to a few seconds each."""
import numpy as np
import pytest
import torch

from oracle import codec_oracle as CO
from oracle import dit_oracle as O
from smalltts_amd.weights import (DEFAULT_CODEC, codec_decoder_param_specs, codec_encoder_param_specs, synth_state_dict)
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300))


@pytest.fixture(scope="module")
def eng(golden_seed):
    from smalltts_amd.engine import HipEngine
    e = HipEngine(0, "bf16x3")
    e.load_synthetic(golden_seed, parts=("dit", "decoder", "encoder"))
    e.finalize()
    return e


def _inputs(B, N, R, P, seed, ragged=False):
    g = torch.Generator().manual_seed(seed)
    ref = torch.randn(B, R, 64, generator=g)
    ids = torch.randint(1, 198, (B, P), generator=g)
    rl = torch.full((B,), R)
    pm = torch.ones(B, P, dtype=torch.bool)
    mask = torch.ones(B, N, dtype=torch.bool)
    if ragged and B > 1:
        rl[-1] = max(1, R // 3); pm[-1, P // 2:] = False; ids[-1, P // 2:] = 0; mask[-1, (2 * N) // 3:] = False
    noise = torch.randn(4, B, N, 64, generator=g)
    return ref, rl, ids, pm, mask, noise


@pytest.mark.parametrize("B,N,R,P,ragged", [(2, 75, 38, 128, True),    # SURVEY §8 "realistic" secondary point
                                            (1, 225, 64, 198, False),   # 30 s, longest reference / phoneme ranges
                                            (1, 1, 1, 1, False)])       # smallest legal call (duration floor = 1 frame)
def test_sampler_vs_oracle_at_range_edges(eng, dit_weights, B, N, R, P, ragged):
    ref, rl, ids, pm, mask, noise = _inputs(B, N, R, P, seed=100 + N, ragged=ragged)
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, ref, rl, ids, pm)
        ox = O.sample_dmd(dit_weights, oc, pm, mask, noise, 4)
    x = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, noise=noise).cpu().numpy()
    m = mask.numpy()
    err = rel_l2(x[m], ox.numpy()[m])
    assert err < 1e-4, f"B={B} N={N} R={R} P={P}: latent rel L2 {err:.3e}"


@pytest.mark.parametrize("steps", [1, 2, 7])
def test_sampler_other_step_counts_vs_oracle(eng, dit_weights, steps):
    """NUM_STEPS is a constant in the reference (4) but a parameter of the operator: linspace(1, 0, n) incl. n = 1 (t = 1 only)."""
    B, N, R, P = 2, 20, 9, 17
    ref, rl, ids, pm, mask, _ = _inputs(B, N, R, P, seed=300 + steps, ragged=True)
    noise = torch.randn(steps, B, N, 64, generator=torch.Generator().manual_seed(steps))
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, ref, rl, ids, pm)
        keep = []
        ox = O.sample_dmd(dit_weights, oc, pm, mask, noise, steps, keep=keep)
    x, per_step = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, num_steps=steps, noise=noise, return_steps=True)
    m = mask.numpy()
    assert rel_l2(x.cpu().numpy()[m], ox.numpy()[m]) < 1e-4
    for i in range(steps):   # every intermediate x-hat, not just the last
        assert rel_l2(per_step[i].cpu().numpy()[m], keep[i].numpy()[m]) < 1e-4, f"step {i}"


def test_full_spec_decode_10s_vs_oracle(eng, golden_seed):
    """The bench's codec workload per utterance: 75 frames -> 240000 samples through the 344 M-parameter decoder."""
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), golden_seed))
    lat = torch.randn(1, 75, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = CO.decode(wd, lat, DEFAULT_CODEC)
    got = eng.codec_decode(lat).cpu()
    assert tuple(got.shape) == (1, 1, 240000)
    s = snr_db(got.numpy(), ref.numpy())
    assert s > 60.0, f"10 s decode SNR {s:.1f} dB"
    # causality at full scale: a batch of 2 with different content in the tail shares the prefix bit-for-bit
    lat2 = torch.cat([lat, lat.clone()], 0); lat2[1, 40:] += 1.0
    g2 = eng.codec_decode(lat2).cpu()
    assert torch.equal(g2[0], got[0]) and torch.equal(g2[1, :, :40 * 3200], got[0, :, :40 * 3200])


def test_full_spec_decode_bench_batch_vs_oracle(eng, golden_seed):
    """The bench's codec workload exactly: 8 x 75 frames.  At this size the coarse stages take the paths a single utterance
    never sees (M = 600 / 4800: 160x128 one-round tiles, split-K second product + RowMap reduce, gemm3 upsampling on the
    split image), so it gets its own check: two utterances against the CPU oracle, all eight against the single-utterance
    GPU result (different tile shapes -> different summation order, hence a bound instead of bit equality)."""
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), golden_seed))
    lat = torch.randn(8, 75, 64, generator=torch.Generator().manual_seed(11))
    got = eng.codec_decode(lat).cpu()
    assert tuple(got.shape) == (8, 1, 240000)
    with torch.no_grad():
        for b in (0, 5):
            ref = CO.decode(wd, lat[b:b + 1], DEFAULT_CODEC)
            s = snr_db(got[b:b + 1].numpy(), ref.numpy())
            assert s > 60.0, f"utterance {b} of the batch of 8: decode SNR {s:.1f} dB"
    for b in range(8):
        one = eng.codec_decode(lat[b:b + 1]).cpu()
        s = snr_db(got[b:b + 1].numpy(), one.numpy())
        assert s > 90.0, f"utterance {b}: batch-of-8 vs single SNR {s:.1f} dB"


@pytest.mark.parametrize("B,T", [(1, 1), (3, 2), (2, 5)])
def test_full_spec_decode_odd_frame_counts_vs_oracle(eng, golden_seed, B, T):
    """Partial tiles of the fused FFN kernels: at T = 1 the C = 256 stage has 200 frames (1.56 passes of 128), the
    C = 32 stage 3200; B = 3, T = 2 and B = 2, T = 5 leave ragged last waves / passes at every stage."""
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), golden_seed))
    lat = torch.randn(B, T, 64, generator=torch.Generator().manual_seed(40 + T))
    with torch.no_grad():
        ref = CO.decode(wd, lat, DEFAULT_CODEC)
    got = eng.codec_decode(lat).cpu()
    assert tuple(got.shape) == (B, 1, 3200 * T)
    s = snr_db(got.numpy(), ref.numpy())
    assert s > 60.0, f"B={B} T={T}: decode SNR {s:.1f} dB"
    one = eng.codec_decode(lat[:1]).cpu()
    assert torch.equal(one[0], got[0])          # batch invariance with ragged tiles


def test_codec_is_bitwise_repeatable_and_batch_invariant(eng):
    """Races in the fused LDS kernels show up as run-to-run or batch-size dependent bits (a missing lgkmcnt drain
    before a raw s_barrier did exactly that)."""
    lat = torch.randn(1, 75, 64, generator=torch.Generator().manual_seed(6))
    runs = [eng.codec_decode(lat).cpu() for _ in range(4)]
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    both = eng.codec_decode(torch.cat([lat, lat], 0)).cpu()
    assert torch.equal(both[0], runs[0][0]) and torch.equal(both[1], runs[0][0])
    au = torch.randn(2, 1, 3200 * 4, generator=torch.Generator().manual_seed(7)) * 0.3
    e = [eng.codec_encode(au).cpu() for _ in range(3)]
    assert all(torch.equal(e[0], r) for r in e[1:])


def test_full_spec_encode_2s_vs_oracle(eng, golden_seed):
    """clone path input: the bench.rs reference clip (2 s, 440 Hz sine) -> 15 latent frames."""
    we = O.to_torch(synth_state_dict(codec_encoder_param_specs(DEFAULT_CODEC), golden_seed))
    t = torch.arange(48000, dtype=torch.float32) / 24000.0
    audio = torch.sin(2 * np.pi * 440.0 * t)[None, None]
    with torch.no_grad():
        ref = CO.encode(we, audio, DEFAULT_CODEC)
    got = eng.codec_encode(audio).cpu()
    assert tuple(got.shape) == (1, 15, 64)
    s = snr_db(got.numpy(), ref.numpy())
    assert s > 60.0, f"encode SNR {s:.1f} dB"


def test_full_spec_encode_bench_batch_vs_oracle(eng, golden_seed):
    """The clone workload's codec encode exactly: 8 reference clips of 2 s (bench.rs:5) in one batch."""
    we = O.to_torch(synth_state_dict(codec_encoder_param_specs(DEFAULT_CODEC), golden_seed))
    t = torch.arange(48000, dtype=torch.float32) / 24000.0
    g = torch.Generator().manual_seed(12)
    audio = torch.stack([torch.sin(2 * np.pi * (220.0 + 55.0 * i) * t) * 0.8 + 0.05 * torch.randn(48000, generator=g)
                         for i in range(8)])[:, None]
    got = eng.codec_encode(audio).cpu()
    assert tuple(got.shape) == (8, 15, 64)
    with torch.no_grad():
        ref = CO.encode(we, audio[3:4], DEFAULT_CODEC)
    s = snr_db(got[3:4].numpy(), ref.numpy())
    assert s > 60.0, f"utterance 3 of the batch of 8: encode SNR {s:.1f} dB"
    for b in range(8):
        one = eng.codec_encode(audio[b:b + 1]).cpu()
        s = snr_db(got[b:b + 1].numpy(), one.numpy())
        assert s > 90.0, f"utterance {b}: batch-of-8 vs single encode SNR {s:.1f} dB"


def test_batch_of_eight_equals_eight_singles_through_the_whole_path(eng):
    """BASELINE 'batch=8' semantics: the reference runs 8 sequential batch-1 calls (bench.rs:42-44); the batched GPU
    path must give each utterance the same audio."""
    ref, rl, ids, pm, mask, noise = _inputs(8, 20, 6, 9, seed=9)
    xb = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, noise=noise)
    ab = eng.codec_decode(xb).cpu()
    for b in (0, 3, 7):
        x1 = eng.sample(eng.cond_encode(ref[b:b + 1], rl[b:b + 1], ids[b:b + 1], pm[b:b + 1]), mask[b:b + 1],
                        noise=noise[:, b:b + 1].contiguous())
        a1 = eng.codec_decode(x1).cpu()
        assert snr_db(ab[b].numpy(), a1[0].numpy()) > 80.0
