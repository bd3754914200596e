"""CPU: structural checks of the build-defined codec oracle (parity vs the reference is unpinned —
see oracle/codec_oracle.py). Pins what the reference does pin: I/O shapes and hop = 3200."""
import numpy as np
import torch

from oracle import codec_oracle as CO
from oracle.dit_oracle import to_torch
from smalltts_amd.weights import (DEFAULT_CODEC, CodecSpec, codec_decoder_param_specs, codec_encoder_param_specs,
                                  synth_state_dict)

SMALL = CodecSpec(latent_dim=64, n_filters=8, ratios=(4, 2, 2), dec_depths=(2, 1, 1, 2))


def test_default_spec_matches_reference_contract():
    assert DEFAULT_CODEC.hop == 3200 and DEFAULT_CODEC.latent_dim == 64  # infer/onnx.py:11-12
    n = sum(int(np.prod(s)) for _, s in codec_decoder_param_specs())
    assert 3.3e8 < n < 3.6e8  # ~340 M parameters per codec half


def test_decode_shape_and_causality():
    w = to_torch(synth_state_dict(codec_decoder_param_specs(SMALL), 1))
    lat = torch.randn(2, 6, 64)
    with torch.no_grad():
        a = CO.decode(w, lat, SMALL)
        lat2 = lat.clone(); lat2[:, 4:] += 1.0
        b = CO.decode(w, lat2, SMALL)
    assert a.shape == (2, 1, SMALL.hop * 6)
    cut = 4 * SMALL.hop
    assert torch.equal(a[..., :cut], b[..., :cut]) and not torch.equal(a[..., cut:], b[..., cut:])


def test_encode_shape_and_causality():
    w = to_torch(synth_state_dict(codec_encoder_param_specs(SMALL), 1))
    au = torch.randn(2, 1, SMALL.hop * 5 + 3)
    with torch.no_grad():
        a = CO.encode(w, au, SMALL)
        au2 = au.clone(); au2[..., 3 * SMALL.hop:] += 0.5
        b = CO.encode(w, au2, SMALL)
    assert a.shape == (2, 5, 64)
    assert torch.equal(a[:, :3], b[:, :3]) and not torch.equal(a[:, 3:], b[:, 3:])
