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


@pytest.mark.parametrize("sr,target", [(16000, 24000), (44100, 24000), (48000, 16000)])
def test_host_resampler_matches_oracle(sr, target):
    """The host resampler of the CLI path against the closed-form float64 oracle (oracle/resample_oracle.py)."""
    from oracle.resample_oracle import resample as oracle_resample
    rng = np.random.default_rng(sr)
    x = (0.4 * np.sin(2 * np.pi * 523.25 * np.arange(int(0.1 * sr)) / sr) + 0.1 * rng.standard_normal(int(0.1 * sr))).astype(np.float32)
    want = oracle_resample(x, sr, target)
    got = audio.resample_hq(x, sr, target)
    assert got.shape == want.shape
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 3e-6


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


def test_checkpoint_converter_roundtrip_and_rejects_wrong_inventory(tmp_path):
    """checkpoint["student_model"] with wrapper prefixes -> weight file holding exactly the DiT inventory."""
    import torch
    from smalltts_amd import convert
    from smalltts_amd.weights import dit_param_specs, load_weight_file
    rng = np.random.default_rng(3)
    specs = dit_param_specs()
    small = {n: rng.standard_normal(s).astype(np.float32) for n, s in specs if int(np.prod(s)) <= 4096}
    # big tensors as zeros-strided views would still be written in full: keep the file small by checking a subset + inventory logic
    sd = {"_orig_mod.module." + n: torch.from_numpy(a) for n, a in small.items()}
    sd["initted"] = torch.tensor(1.0)
    ck = tmp_path / "ck.pt"
    torch.save({"student_model": sd, "step": 7}, ck)
    out = tmp_path / "w.smtts"
    with pytest.raises(ValueError, match="not a DiTModel"):
        convert.convert_checkpoint(str(ck), str(out))                      # most of the inventory is missing
    rep = convert.convert_checkpoint(str(ck), str(out), allow_partial=True)
    assert rep["matched"] == len(small) and not rep["shape_mismatch"] and not rep["unexpected"]
    assert len(rep["missing"]) == len(specs) - len(small)
    got, codec = load_weight_file(str(out))
    assert codec is None and set(got) == set(small)
    assert all(np.array_equal(got[n], small[n]) for n in small)


def _pb_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_field(no, wt, payload):
    return _pb_varint(no << 3 | wt) + (_pb_varint(len(payload)) + payload if wt == 2 else payload)


def test_onnx_initializer_reader_on_hand_encoded_model(tmp_path):
    """The wire-format reader needs no onnx package: raw_data fp32, packed float_data, fp16 raw, unpacked dims."""
    from smalltts_amd import convert
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.array([0.5, -1.25], dtype=np.float32)
    c = np.array([[1.0, 2.0]], dtype=np.float16)
    t_a = (_pb_field(1, 2, _pb_varint(2) + _pb_varint(3)) + _pb_field(2, 0, _pb_varint(1)) +
           _pb_field(8, 2, b"velocity.weight") + _pb_field(9, 2, a.tobytes()))
    t_b = (_pb_field(1, 0, _pb_varint(2)) + _pb_field(2, 0, _pb_varint(1)) + _pb_field(4, 2, b.tobytes()) +
           _pb_field(8, 2, b"velocity.bias"))
    t_c = (_pb_field(1, 2, _pb_varint(1) + _pb_varint(2)) + _pb_field(2, 0, _pb_varint(10)) + _pb_field(8, 2, b"h") +
           _pb_field(9, 2, c.tobytes()))
    graph = _pb_field(1, 2, b"node-bytes-ignored") + _pb_field(5, 2, t_a) + _pb_field(5, 2, t_b) + _pb_field(5, 2, t_c)
    model = _pb_field(1, 0, _pb_varint(8)) + _pb_field(2, 2, b"pytorch") + _pb_field(7, 2, graph)
    p = tmp_path / "m.onnx"
    p.write_bytes(model)
    got = convert.read_onnx_initializers(str(p))
    assert set(got) == {"velocity.weight", "velocity.bias", "h"}
    assert np.array_equal(got["velocity.weight"], a) and np.array_equal(got["velocity.bias"], b)
    assert got["h"].dtype == np.float16 and np.array_equal(got["h"], c)
    # name matching: the two velocity tensors have the wrong shapes for this build -> reported, nothing guessed
    with pytest.raises(ValueError, match="do not cover"):
        convert.convert_onnx([str(p)], str(tmp_path / "o.smtts"))


def test_text_normalizer_spells_out_what_the_reference_spells_out():
    """Reference _phonemize runs EnglishTextNormalizer.normalize before espeak (phonemes.py:67-70); sentences from its own
    __main__ list (phonemes.py:121-145).  Expected strings follow normalizer.py's rules with inflect's documented spelling
    (inflect is not installed offline: parity with inflect itself is unpinned, see smalltts_amd/normalizer.py)."""
    from smalltts_amd.normalizer import EnglishTextNormalizer, needs_normalization
    n = EnglishTextNormalizer()
    sq = lambda t: " ".join(n.normalize(t).split())
    assert sq("Dr. Smith and Mrs. Johnson met at 3:30pm.") == "doctor. Smith and misess. Johnson met at three : thirty pm."
    assert sq("The company earned $1,250,000.50 in Q4 2023.") == ("The company earned one million, two hundred fifty thousand "
                                                                    "dollars, fifty cents in Q four twenty twenty-three .")
    assert sq("About 75% of students scored above 90th percentile.") == "About seventy-five percent of students scored above ninetieth percentile."
    assert sq("The recipe calls for 1/2 cup sugar and 3/4 tsp salt.") == "The recipe calls for one half cup sugar and three quarters tsp salt."
    assert sq("The temperature is 98.6°F today.") == "The temperature is ninety-eight point six °F today."
    assert sq("£500 equals approximately $625.50.") == "five hundred pounds equals approximately six hundred twenty-five dollars, fifty cents ."
    assert sq("The 21st century began on January 1st, 2001.") == "The twenty-first century began on January first , two thousand one ."
    assert sq("BTW, the meeting is at 2nd St. near Fort Collins.") == "by the way, the meeting is at second saint. near Fort Collins."
    assert sq("1905 1100 2000 1984 7/8 101st $1 $0.01") == ("nineteen oh five eleven hundred two thousand nineteen eighty-four seven eighth "
                                                            "one hundred and first one dollar one cent")
    assert needs_normalization("at 3pm") and needs_normalization("Mr. X") and not needs_normalization("hello there, world!")
    # the token path normalises before it maps characters (chars backend: no espeak offline)
    assert phonemes.get_token_ids("$5", backend="chars") == [phonemes.p2idx[c] for c in n.normalize("$5") if c in phonemes.p2idx]
    assert phonemes.decode_token_ids(phonemes.get_token_ids("$5", backend="chars")).split() == ["five", "dollars"]


def test_reference_import_paths_resolve_to_this_build():
    """The reference's scripts import smalltts.* (tryme.py:7-9, clone.py:7-11, interactive.py:11-15); the shim package re-exports
    this build under those paths so the import lines need no edit.  (No GPU here: classes are imported, not constructed.)"""
    import smalltts
    from smalltts.assets.ensure import ensure_assets
    from smalltts.codec.onnx import Decoder, Encoder
    from smalltts.data.phonemization.phonemes import get_token_ids, phoneme_len
    from smalltts.infer.onnx import HOP_SIZE, NUM_STEPS, SAMPLE_RATE, SmallTTS, estimate_duration
    from smalltts.infer.utils import resample_hq
    from smalltts_amd import api
    assert smalltts.SmallTTS is SmallTTS is api.SmallTTS and Encoder is api.Encoder and Decoder is api.Decoder
    assert (SAMPLE_RATE, HOP_SIZE, NUM_STEPS, phoneme_len) == (24000, 3200, 4, 198)
    assert estimate_duration("x" * 23) == 2.0 and estimate_duration("") == 0.5 and estimate_duration("x" * 1000) == 30.0
    import torch
    assert tuple(resample_hq(torch.zeros(1, 1600), 16000, 24000).shape) == (1, 2400)
    assert get_token_ids("[laughter]", backend="chars") == [188] * 4
    with pytest.raises(FileNotFoundError, match="smalltts_amd.convert"):
        ensure_assets(["codec", "dmd"])
    with pytest.raises(AttributeError):
        smalltts.nothing_here


def _onnx_bytes(named):
    """Hand-encoded ModelProto whose graph holds the given float32 initialisers (name, array) in order."""
    graph = b""
    for name, arr in named:
        dims = b"".join(_pb_varint(int(d)) for d in arr.shape)
        t = (_pb_field(1, 2, dims) if arr.ndim else b"") + _pb_field(2, 0, _pb_varint(1)) + _pb_field(8, 2, name.encode()) + \
            _pb_field(9, 2, np.ascontiguousarray(arr, "<f4").tobytes())
        graph += _pb_field(5, 2, t)
    return _pb_field(1, 0, _pb_varint(8)) + _pb_field(7, 2, graph)


def test_codec_onnx_with_renamed_and_permuted_initialisers_gets_a_precise_report(tmp_path):
    """the day decoder.onnx exists, pinning the codec must be one command.  A hand-built ONNX file holds a
    small codec decoder the way an exporter would leave it — Linear weights as anonymous transposed MatMul operands, conv
    weights under graph names, depthwise weights as (C, 1, K), one bias family missing, one extra tensor — and the converter
    (a) reports the deltas per shape signature, (b) maps the rest by position, (c) writes a weight file the loader accepts."""
    from smalltts_amd import convert
    from smalltts_amd.weights import CodecSpec, codec_decoder_param_specs, load_weight_file, synth_state_dict
    spec = CodecSpec(n_filters=8, ratios=(4, 2), dec_depths=(1, 2, 1))
    sd = synth_state_dict(codec_decoder_param_specs(spec), 5)
    named, k = [], 0
    for name, arr in sd.items():
        if name.endswith(".ffn.w1.bias"):
            continue                                            # the exporter's graph has no FFN-1 bias
        if name.endswith(".ffn.w1.weight") or name.endswith(".ffn.w2.weight"):
            named.append((f"onnx::MatMul_{1000 + k}", arr.T)); k += 1          # anonymous, stored [in, out]
        elif name.endswith(".mixer.weight"):
            named.append((f"/decoder/stages/{k}/mixer/conv.weight", arr[:, None, :])); k += 1   # torch depthwise layout (C, 1, K)
        elif name.endswith((".up.1.weight", ".up.2.weight", ".stem.weight", ".head.weight")):
            named.append((name, arr))                           # kept their names
        else:
            named.append((f"decoder.renamed.{k}", arr)); k += 1
    named.append(("decoder.extra_buffer", np.zeros((3, 5, 7), np.float32)))
    p = tmp_path / "decoder.onnx"
    p.write_bytes(_onnx_bytes(named))
    out = tmp_path / "codec.smtts"
    with pytest.raises(ValueError, match="--map-by-position"):
        convert.convert_onnx([str(p)], str(out), codec=spec, parts=["decoder"])
    rep = convert.convert_onnx([str(p)], str(out), codec=spec, allow_partial=True, map_by_position=True, parts=["decoder"])
    sg = rep["signature"]
    assert set(sg["by_name"]) == {n for n in sd if n.endswith((".up.1.weight", ".up.2.weight", ".stem.weight", ".head.weight"))}
    # every FFN weight came back under its own name, un-transposed, purely from graph order within its shape signature
    got, cdict = load_weight_file(str(out))
    for n, a in sd.items():
        if n.endswith((".ffn.w1.weight", ".ffn.w2.weight", ".mixer.weight")):
            assert np.array_equal(got[n], a), n
    # ... and the report names exactly what differs: the missing bias family and the extra tensor
    missing = [n for n in sd if n.endswith(".ffn.w1.bias")]
    assert set(missing) <= set(rep["missing"])
    deltas = sg["count_mismatch"] + sg["expected_only"]
    assert any(set(row.get("expected_names", row.get("names"))) & set(missing) for row in deltas), sg
    assert any(row["names"] == ["decoder.extra_buffer"] for row in sg["onnx_only"])
    assert cdict["ratios"] == [4, 2] and cdict["ffn_bias"] is True
    # the CLI prints the same report and exits non-zero without --allow-partial
    assert convert.main(["--onnx", str(p), "--codec-spec", _write_json(tmp_path, spec.to_dict()), "--map-by-position", "--parts", "decoder", "--out", str(out),
                         "--allow-partial", "--report", str(tmp_path / "r.json")]) == 0
    import json
    assert json.load(open(tmp_path / "r.json"))["signature"]["summary"].startswith(f"{len(sg['by_name'])} by name")


def _write_json(tmp_path, obj):
    import json
    p = tmp_path / "spec.json"
    p.write_text(json.dumps(obj))
    return str(p)


def test_codec_spec_flags_steer_the_inventory():
    from smalltts_amd.weights import CodecSpec, codec_decoder_param_specs, codec_encoder_param_specs
    full = CodecSpec(n_filters=8, ratios=(4, 2), dec_depths=(1, 1, 1), final_norm=True)
    lean = CodecSpec(n_filters=8, ratios=(4, 2), dec_depths=(1, 1, 1), conv_bias=False, ffn_bias=False, layer_scale=False)
    nf = {n for n, _ in codec_decoder_param_specs(full) + codec_encoder_param_specs(full)}
    nl = {n for n, _ in codec_decoder_param_specs(lean) + codec_encoder_param_specs(lean)}
    assert "codec.decoder.final_norm.weight" in nf and "codec.encoder.final_norm.weight" in nf
    assert not any(n.endswith((".bias", "gamma")) or "final_norm" in n for n in nl)
    assert nl < nf and CodecSpec(**lean.to_dict()).to_dict() == lean.to_dict()


def test_gelu_q5_constants_in_the_kernel_header_meet_their_error_bound():
    """GeluQ5 (csrc/common.hpp): gelu(x) = max(x, 0) - |x| 2^q(|x|).  The constants compiled into the kernels are parsed from the
    header and evaluated in fp32 exactly as the device does; the bound is the one the header states (2.1e-6 absolute, all x)."""
    import os
    import re
    from tests.studies.gelu_q5_fit import fp32_error
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smalltts_amd", "csrc", "common.hpp")).read()
    body = re.search(r"struct GeluQ5 \{(.*?)\};", src, re.S).group(1)
    c = np.array([float(v) for v in re.findall(r"Q\d = (-?[0-9.e-]+)f", body)], np.float32)
    assert c.size == 6 and c[-1] < 0            # negative leading coefficient: |x| -> inf extrapolates to max(x, 0)
    assert fp32_error(c) < 2.2e-6


def test_packed_fp16_gelu_of_the_default_degree_is_relatively_accurate_at_small_activations():
    """the packed-fp16 GELU of the fused codec FFN kernels (csrc/common.hpp gelu_q5_pk_*) must be accurate RELATIVE to the
    value for small activations too — a fit whose 2^Q0 is not ~0.5 biases every small hidden value by a fixed fraction (degree 3:
    +0.29 %).  The constants of the degree the header selects are parsed from it and evaluated with every step rounded to fp16, as
    the device's v_pk_fma_f16 chain does (tests/studies/gelu_f16_packed.py)."""
    import os
    import re
    import torch
    from tests.studies.gelu_f16_packed import gelu_pk_f16_deg
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smalltts_amd", "csrc", "common.hpp")).read()
    deg = int(re.search(r"#ifndef GELU_PK_DEG\n#define GELU_PK_DEG (\d)", src).group(1))
    body = re.search(r"struct GeluQ%d \{(.*?)\};" % deg, src, re.S).group(1)
    c = [float(v) for v in re.findall(r"Q\d = (-?[0-9.e-]+)f", body)]
    assert len(c) == deg + 1 and abs(2.0 ** c[0] - 0.5) < 5e-4          # gelu(x) -> x / 2 at small |x|: no fixed-fraction bias
    assert c[-1] < 0     # negative leading coefficient: 2^q -> 0 for large |x| (degree 4 ends positive: 2^q overflows past |x| ~ 20, NaN audio)
    big = torch.tensor([30.0, -30.0, 800.0, -800.0, 60000.0, -60000.0])
    assert torch.equal(gelu_pk_f16_deg(c)(big), torch.clamp_min(big.to(torch.float16).float(), 0))
    g = gelu_pk_f16_deg(c)
    x = torch.linspace(-12, 12, 400001)
    exact = 0.5 * x.double() * (1 + torch.erf(x.double() / np.sqrt(2)))
    err = (g(x).double() - exact).abs()
    small = (x.abs() < 0.25) & (x.abs() > 1e-3)
    assert float((err[small] / exact[small].abs()).max()) < 2.5e-3       # degree 3 reads 7.2e-3 here, degree 5 1.3e-3
    assert float((err / exact.abs().clamp_min(1e-3)).max()) < 1.2e-2      # degree 3: 2.3e-2
    xn = torch.randn(1000000, generator=torch.Generator().manual_seed(0)) * 1.5
    en = 0.5 * xn.double() * (1 + torch.erf(xn.double() / np.sqrt(2)))
    assert float(((g(xn).double() - en) ** 2).mean().sqrt()) < 3.2e-4    # the fp16 Horner steps' own floor is 3.1e-4


def test_optional_codec_tensor_switched_off_by_the_spec_is_an_error_not_a_silent_drop():
    """_validated() filters a source down to the inventory of the CodecSpec in force; a final-norm weight or a bias
    that the spec says is absent would have vanished without a word although the engine applies such tensors when present."""
    import numpy as np
    import pytest
    from smalltts_amd.api import _validated
    from smalltts_amd.weights import CodecSpec, all_param_specs
    on = CodecSpec(final_norm=True)
    off = CodecSpec(final_norm=False, ffn_bias=False)
    tensors = {k: np.zeros(s, np.float32) for k, s in all_param_specs(on) if k.startswith("codec.decoder.")}
    assert len(_validated(tensors, "t", on)) == len(tensors)
    with pytest.raises(ValueError, match="optional codec parameter"):
        _validated(tensors, "t", off)
    kept = {k: v for k, v in tensors.items() if k in dict(all_param_specs(off))}
    assert 0 < len(_validated(kept, "t", off)) == len(kept) < len(tensors)
    extra = dict(kept, **{"optimizer.state.step": np.zeros(1, np.float32)})      # unrelated keys are still just ignored
    assert len(_validated(extra, "t", off)) == len(kept)


def test_bench_codec_bytes_follow_the_spec_and_profiler_names_map_back():
    """SURVEY 8(d) bytes of the codec decode are derived from the CodecSpec in code (2 x stage-boundary images + weights at 2 B), and
    the rocprofv3 kernel names of the kernels bench.py may crown as `roofline.kernel` map back to the in-process profiler's class
    names — otherwise the stamped averages / traffic under profiles/ could not be quoted for them."""
    import importlib, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
    bench = importlib.import_module("bench")
    from smalltts_amd.weights import DEFAULT_CODEC, CodecSpec
    b = bench.codec_decode_algo_bytes()
    imgs = sum(8 * 75 * int(np.prod(DEFAULT_CODEC.ratios[:i])) * (32 << (6 - i)) * 4 for i in range(7)) + 8 * 240000 * 4
    assert 2.55e9 < b < 2.59e9 and b > 2 * imgs and (b - 2 * imgs) % 2 == 0           # the rest = weights at 2 B / parameter
    small = bench.codec_decode_algo_bytes(CodecSpec(n_filters=8, dec_depths=(1,) * 7), batch=1, frames=2)
    assert 0 < small < b / 100
    from prof_names import prof_name
    for k, want in {
        "void gemm3_kernel<64, 64, 2, 2, 2, 4, EpiResid<1> >(Gemm3Operands, EpiResid<1>)": "gemm3<64x64,s2,resid_gate>",
        "void gemm4_kernel<2, EpiStore<2> >(Gemm3Operands, EpiStore<2>)": "gemm4<256x256,s2,store_gelu>",
        "void gemm3_kernel<64, 64, 2, 2, 2, 4, EpiResidLN>(Gemm3Operands, EpiResidLN)": "gemm3<64x64,s2,resid_ln>",
        "void gemm3_kernel<128, 128, 4, 2, 2, 4, EpiSwiGLUT<true> >(Gemm3Operands, EpiSwiGLUT<true>)": "gemm3<128x128,s2,swiglu>",
        "void gemm3_kernel<64, 128, 2, 4, 2, 3, EpiQKVT<false> >(Gemm3Operands, EpiQKVT<false>)": "gemm3<64x128,s2,qkv_img>",
        "void codec_chain_wave_kernel<32, 2, 12, 3>(FfnChainArgs)": "codec_chain_wave<32>",
        "void codec_ffn_wave_kernel<64, 2, 8, true>(FfnWaveArgs)": "codec_block_wave<64>",
        "void codec_ffn_wave_kernel<32, 2, 8, false>(FfnWaveArgs)": "codec_ffn_wave<32>",
        "void codec_ffn_stream_kernel<128, 2, 8, 4>(FfnStreamArgs)": "codec_ffn_stream<128>",
        "void mixer_stream_kernel<128>(MixerArgs)": "mixer_fused",
        "void (anonymous namespace)::attention_img_kernel<128, 2>(AttnImg, int)": "attention_img<128>",
    }.items():
        assert prof_name(k) == want, (k, prof_name(k))


def test_the_shipped_library_reads_exactly_the_ten_documented_environment_switches():
    """every getenv("SMTTS_*") left in csrc/ is one of the ten product switches DESIGN.md lists; the A/B sessions'
    switches go through lab_env(), which sees the environment in a -DSMTTS_LAB build only."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for f in glob.glob(os.path.join(root, "smalltts_amd", "csrc", "*.h*")):
        seen |= set(re.findall(r'(?<!lab_)getenv\("(SMTTS_[A-Z0-9_]+)"\)', open(f).read()))
    want = {"SMTTS_ATTN_EPI", "SMTTS_GEMM_DEEP", "SMTTS_GEMM_GROUP", "SMTTS_GEMM_XCD", "SMTTS_LN_FOLD", "SMTTS_MIXER_STREAM",
            "SMTTS_MIXER_WIDE", "SMTTS_PERSIST_CUS", "SMTTS_SINGLE_STREAM", "SMTTS_STAGE_CHAIN"}
    assert seen == want, (seen - want, want - seen)
    design = open(os.path.join(root, "DESIGN.md")).read()
    assert all(name in design for name in want)
    # no experiment branch or timeline stamp in the shipped streamed-FFN kernel: those live in the lab copy
    shipped = open(os.path.join(root, "smalltts_amd", "csrc", "codec_ffn_stream.hip")).read()
    assert not re.search(r"#\s*if(n?def)?\s+FS_(ELIM|LIN_STORE|NOBARRIER|TIMELINE|PHASE_TICKS|EPI_NOFENCE)", shipped) and "FS_STAMP" not in shipped
    lab = os.path.join(root, "smalltts_amd", "csrc", "exp", "codec_ffn_stream_lab.hip")
    assert os.path.exists(lab)
    # ... and the shipped kernel IS the lab copy with the experiment switches resolved as "not defined": the two cannot drift apart
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    from strip_lab import STREAM_FFN, shipped as strip_to_shipped
    with open(lab) as f:
        assert "".join(strip_to_shipped(f.readlines(), STREAM_FFN)) == shipped
