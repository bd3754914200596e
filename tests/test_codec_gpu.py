"""GPU: codec decode/encode through the C ABI vs the CPU codec oracle (same seeded weights).
Bound stated by the build: SNR >= 60 dB for the split-bf16 path (parity vs the reference's ONNX codec
is unpinned — no codec source or weights exist in the reference tree)."""
import numpy as np
import pytest
import torch

from oracle import codec_oracle as CO
from oracle.dit_oracle import to_torch
from smalltts_amd.weights import (DEFAULT_CODEC, CodecSpec, codec_decoder_param_specs, codec_encoder_param_specs,
                                  synth_state_dict)

pytestmark = pytest.mark.gpu
SNR_BOUND_DB = 60.0


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300))


SPECS = {
    "tiny": CodecSpec(n_filters=8, ratios=(4, 2, 2), dec_depths=(2, 1, 1, 2)),
    "odd": CodecSpec(n_filters=16, ratios=(5, 3, 2), dec_depths=(1, 2, 1, 1)),
    "narrow6": CodecSpec(n_filters=32, ratios=(8, 5, 5, 4, 2, 2), dec_depths=(1, 1, 1, 1, 1, 1, 1)),
    # the plausible deltas of an exported VibeVoice codec: no conv / FFN biases, no layer scale, a final
    # norm in front of the head, a different depth order — absent tensors take their identity value in the engine and the oracle
    "no_bias_no_scale_final_norm": CodecSpec(n_filters=32, ratios=(8, 5, 5, 4, 2, 2), dec_depths=(1, 2, 1, 1, 1, 1, 2), conv_bias=False,
                                             ffn_bias=False, layer_scale=False, final_norm=True),
    "ffn_bias_only": CodecSpec(n_filters=16, ratios=(5, 3, 2), dec_depths=(1, 2, 1, 1), conv_bias=False, layer_scale=True, final_norm=True),
}


@pytest.mark.parametrize("name", list(SPECS))
def test_decode_and_encode_vs_oracle(name):
    from smalltts_amd.engine import HipEngine
    spec = SPECS[name]
    eng = HipEngine(0, "bf16x3")
    eng.load_synthetic(3, parts=("decoder", "encoder"), codec_spec=spec)
    eng.finalize()
    assert eng.has("decoder") and eng.has("encoder") and eng.hop == spec.hop
    wd = to_torch(synth_state_dict(codec_decoder_param_specs(spec), 3))
    we = to_torch(synth_state_dict(codec_encoder_param_specs(spec), 3))
    g = torch.Generator().manual_seed(0)
    B, T = (3, 7) if spec.hop < 1000 else (2, 3)
    lat = torch.randn(B, T, 64, generator=g)
    with torch.no_grad():
        ref = CO.decode(wd, lat, spec)
    got = eng.codec_decode(lat).cpu()
    assert got.shape == ref.shape == (B, 1, spec.hop * T)
    s = snr_db(got.numpy(), ref.numpy())
    assert s > SNR_BOUND_DB, f"decode SNR {s:.1f} dB"
    audio = torch.randn(B, 1, spec.hop * T + 5, generator=g) * 0.3
    with torch.no_grad():
        rl = CO.encode(we, audio, spec)
    gl = eng.codec_encode(audio).cpu()
    assert gl.shape == rl.shape == (B, T, 64)
    s = snr_db(gl.numpy(), rl.numpy())
    assert s > SNR_BOUND_DB, f"encode SNR {s:.1f} dB"


def test_full_spec_decode_vs_oracle():
    """The real (VibeVoice-shaped, ~344 M parameter) decoder: 2 utterances x 4 frames."""
    from smalltts_amd.engine import HipEngine
    spec = DEFAULT_CODEC
    eng = HipEngine(0, "bf16x3")
    eng.load_synthetic(9, parts=("decoder",), codec_spec=spec)
    eng.finalize()
    wd = to_torch(synth_state_dict(codec_decoder_param_specs(spec), 9))
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 4, 64, generator=g)
    with torch.no_grad():
        ref = CO.decode(wd, lat, spec)
    got = eng.codec_decode(lat).cpu()
    assert got.shape == (2, 1, 3200 * 4)
    s = snr_db(got.numpy(), ref.numpy())
    eng.set_precision("bf16")
    s1 = snr_db(eng.codec_decode(lat).cpu().numpy(), ref.numpy())
    eng.set_precision("bf16x3")
    eng.lib.smtts_test_set_fused_ffn(eng.h, 0)
    s2 = snr_db(eng.codec_decode(lat).cpu().numpy(), ref.numpy())
    eng.lib.smtts_test_set_fused_ffn(eng.h, 1)
    print(f"\n[codec] decode SNR split-bf16 {s:.1f} dB (unfused FFN path {s2:.1f} dB) ; single-pass bf16 {s1:.1f} dB "
          f"(bound {SNR_BOUND_DB} dB)")
    assert s2 > SNR_BOUND_DB
    assert s > SNR_BOUND_DB, f"decode SNR {s:.1f} dB"


_ALT = r"""
import sys, numpy as np, torch
from smalltts_amd.engine import HipEngine
eng = HipEngine(0)
eng.load_synthetic(5, parts=("decoder",)); eng.finalize()
eng.set_tuning(sys.argv[2])
lat = torch.randn(2, 75, 64, generator=torch.Generator().manual_seed(21)).cuda()
np.save(sys.argv[1], eng.codec_decode(lat).cpu().numpy())
"""


def _decode_in_subprocess(tmp_path, tag, env_extra, tuning="latency"):
    """The A/B switches are read when the engine is created: each alternative runs in its own interpreter."""
    import os, subprocess, sys
    out = tmp_path / f"{tag}.npy"
    env = dict(os.environ, **env_extra)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _ALT, str(out), tuning], check=True, env=env, timeout=600)
    return np.load(out)


def test_alternative_paths_agree_with_the_default(tmp_path):
    """Full-size decode (2 x 75 frames) on the A/B alternatives of the round-3 / round-4 changes: scheduling-only switches (tile orders,
    the persistent kernels' grid, ring depth, the streaming form of the C = 128 / 256 mixer and its segment length) must not change a bit;
    the two-kernel wide-stage mixer differs by fp32 summation order."""
    base = _decode_in_subprocess(tmp_path, "base", {})
    for tag, env in (("tile_orders", {"SMTTS_GEMM_XCD": "0", "SMTTS_GEMM_GROUP": "1"}), ("shallow", {"SMTTS_GEMM_DEEP": "0"}),
                     ("staged_mixer", {"SMTTS_MIXER_STREAM": "0"}), ("mixer_segments_of_122", {"SMTTS_MIXER_STREAM": "122"})):
        alt = _decode_in_subprocess(tmp_path, tag, env)
        assert np.array_equal(alt, base), tag
    # round 5, the stage chain (all three blocks of the C = 32 stage per tile in one launch, halos carried in LDS, one warm-up tile per
    # run of tiles): the same arithmetic per frame as one launch per block -> not a bit may change.  Latency tuning: runs of 5 tiles,
    # utterance starts coincide with run starts; throughput tuning (192 workgroups): runs of 7 tiles, the second utterance starts in
    # the MIDDLE of a run (7500 tiles per utterance); 100 workgroups: runs of 13
    # (since round 6 the engine takes the chain only from 16 tiles per wave up — below that the warm-up tile costs what the chain saves,
    # profiles/r06q_chain_small_batch.txt — so at this test's size the default IS one launch per block; SMTTS_STAGE_CHAIN=2 forces the chain)
    assert np.array_equal(_decode_in_subprocess(tmp_path, "no_chain", {"SMTTS_STAGE_CHAIN": "0"}), base), "stage chain off"
    assert np.array_equal(_decode_in_subprocess(tmp_path, "chain", {"SMTTS_STAGE_CHAIN": "2"}), base), "stage chain"
    tp = _decode_in_subprocess(tmp_path, "tp", {}, "throughput")
    assert snr_db(tp, base) > 80.0     # throughput tuning is held to the latency-tuned decode, not only to its own variants
    assert np.array_equal(_decode_in_subprocess(tmp_path, "tp_chain", {"SMTTS_STAGE_CHAIN": "2"}, "throughput"), tp), "stage chain (tp)"
    assert np.array_equal(_decode_in_subprocess(tmp_path, "tp_chain_100", {"SMTTS_STAGE_CHAIN": "2", "SMTTS_PERSIST_CUS": "100"}, "throughput"), tp), "stage chain, 100 workgroups"
    for tag, env in (("grid_all_cus", {"SMTTS_PERSIST_CUS": "0"}), ("grid_half", {"SMTTS_PERSIST_CUS": "128"}),
                     ("tp_shallow", {"SMTTS_GEMM_DEEP": "0"})):
        alt = _decode_in_subprocess(tmp_path, tag, env, "throughput")
        assert np.array_equal(alt, tp), tag
    # the two-kernel mixer sums the rows' squares in another order; at the default precision the normalised rows are then rounded to
    # fp16, so a last-bit difference moves some of those roundings: the two paths differ by about what each differs from fp32
    # (68.7 dB vs the oracle; measured 72 dB between them)
    two = _decode_in_subprocess(tmp_path, "two_kernel_mixer", {"SMTTS_MIXER_WIDE": "0"})
    assert snr_db(two, base) > 66.0
