"""CPU: host-side helpers of the CLI surface (wav I/O, resampler, weight-file format, checkpoint key clean-up,
token helpers, frame/duration arithmetic of the reference)."""
import numpy as np
import pytest

from smalltts_amd import audio, phonemes, weights


def test_wav_roundtrip_pcm16_and_clamp(tmp_path):
    x = np.concatenate([np.linspace(-1.5, 1.5, 1000), [0.0, 1.0, -1.0]]).astype(np.float32)
    p = str(tmp_path / "a.wav")
    audio.write_wav_pcm16(p, x, 24000)
    y, sr = audio.read_wav(p)
    assert sr == 24000 and y.shape == x.shape, (sr, y.shape, x.shape)
    np.testing.assert_allclose(y, np.clip(x, -1, 1), atol=1.6 / 32768)  # scale mismatch 32767/32768 + half an LSB; clamp then * 32767 (audio.rs:22-37)


def test_resampler_identity_length_and_tone():
    x = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000).astype(np.float32)
    assert audio.resample_hq(x, 24000, 24000) is not None and audio.resample_hq(x, 24000, 24000).shape == x.shape
    y = audio.resample_hq(x, 16000, 24000)
    assert y.shape == (24000,)                         # ceil(new * len / orig), torchaudio convention
    ref = np.sin(2 * np.pi * 440 * np.arange(24000) / 24000)
    assert np.abs(y[3000:21000] - ref[3000:21000]).max() < 2e-3
    z = audio.resample_hq(np.stack([x, -x]), 16000, 8000)
    assert z.shape == (2, 8000) and np.allclose(z[0], -z[1], atol=1e-6)
    hi = np.sin(2 * np.pi * 7000 * np.arange(16000) / 16000).astype(np.float32)   # above the new Nyquist: rejected
    assert np.abs(audio.resample_hq(hi, 16000, 8000)[500:-500]).max() < 1e-3


def test_weight_file_roundtrip(tmp_path):
    spec = weights.CodecSpec(n_filters=8, ratios=(4, 2), dec_depths=(1, 1, 1))
    sd = weights.synth_state_dict(weights.codec_decoder_param_specs(spec)[:7] + [("style_encoder.log_scale", ())], 3)
    p = str(tmp_path / "w.smtts")
    weights.save_weight_file(p, sd, spec)
    got, codec = weights.load_weight_file(p)
    assert list(got) == list(sd) and codec["ratios"] == [4, 2]
    for k in sd:
        assert got[k].shape == sd[k].shape and np.array_equal(got[k], sd[k])
    with pytest.raises(ValueError):
        open(p, "r+b").write(b"XXXX")
        weights.load_weight_file(p)


def test_checkpoint_key_cleanup_matches_reference_prefixes():
    sd = {"module._orig_mod.dit.x": 1, "ema_model.online_model.velocity.weight": 2, "a._orig_mod.b": 3, "step": 4,
          "initted": 5}
    assert weights.clean_state_dict_keys(sd) == {"dit.x": 1, "velocity.weight": 2, "a.b": 3}  # distill.py:39-57


def test_synth_recipe_is_order_independent_and_seeded():
    a = weights.synth_tensor("velocity.weight", (64, 960), 1)
    assert np.array_equal(a, weights.synth_tensor("velocity.weight", (64, 960), 1))
    assert not np.array_equal(a, weights.synth_tensor("velocity.weight", (64, 960), 2))
    assert abs(float(a.std()) - 1 / np.sqrt(960)) < 2e-3 and abs(float(a.mean())) < 1e-3
    assert weights.init_rule("dit.transformer_blocks.0.attn.q_norm.weight", (8, 120)) == (1.0, 0.2)
    assert weights.init_rule("dit.transformer_blocks.0.attn_norm.linear.weight", (5760, 960))[0] == 0.0  # a Linear, not a norm


def test_token_helpers_and_event_expansion():
    assert phonemes.parse_tokens_arg("1, 2 3") == [1, 2, 3] and phonemes.parse_tokens_arg("[4,5]") == [4, 5]
    ids = phonemes.get_token_ids("[laughter]", backend="chars")
    assert ids == [phonemes.p2idx["[laughter]"]] * phonemes.NV_REPEAT                  # phonemes.py:39,86-88
    assert phonemes.get_token_ids("[nosuchevent]", backend="chars") == []
    s = "ab, c"
    assert phonemes.decode_token_ids(phonemes.get_token_ids(s, backend="chars")) == s
    assert phonemes.event_id("Whistle") == 197


def test_frame_arithmetic_matches_reference():
    from smalltts_amd.weights import DEFAULT_CODEC
    assert DEFAULT_CODEC.hop == 3200
    for dur, n in ((10.0, 75), (2.0, 15), (0.05, 1), (1.7, 12), (30.0, 225)):   # max(1, int(d * 24000 / 3200)), onnx.py:84
        assert max(1, int(dur * 24000 / 3200)) == n
