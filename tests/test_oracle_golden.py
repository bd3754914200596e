"""Pins oracle/dit_oracle.py to vectors produced by the reference's own PyTorch modules
(tests/golden/make_golden.py). CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dit_oracle as O
from tests.conftest import GOLDEN, golden, rel_l2

TOL = 2e-5  # fp32 vs fp32, different op order (rel L2)


def test_inventory_matches_reference():
    from smalltts_amd.weights import dit_param_specs
    with open(os.path.join(GOLDEN, "state_dict_inventory.json")) as f:
        inv = json.load(f)
    specs = dit_param_specs()
    assert inv["n_tensors"] == len(specs) == 592
    assert inv["n_params"] == sum(int(np.prod(s)) if s else 1 for _, s in specs) == 327_756_609
    assert [[n, list(s)] for n, s in specs] == inv["tensors"]


def test_schedule_kat():
    k = golden("kat_schedule_rope.npz")
    for t, (a, s) in zip(k["ts4"], k["alpha_sigma_4"]):
        oa, os_ = O.alpha_sigma(float(t))
        assert oa == a and os_ == s  # bit-exact: same float64 formula
    for t, (a, s) in zip(k["ts_dense"], k["alpha_sigma_dense"]):
        oa, os_ = O.alpha_sigma(float(t))
        assert abs(float(oa) - float(a)) <= 1e-7 and abs(float(os_) - float(s)) <= 1e-7
    # SURVEY §8a S1a known answers
    a, s = O.alpha_sigma(1.0)
    assert abs(float(a) - 7.853981515e-6) < 1e-12 and float(s) == 1.0
    a, s = O.alpha_sigma(0.0)
    assert float(a) == 1.0 and abs(float(s) - 3.14159297e-5) < 1e-11


def test_rope_kat():
    k = golden("kat_schedule_rope.npz")
    np.testing.assert_allclose(O.rope_angles(8).numpy(), k["rope_8"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.rope_angles(75).numpy(), k["rope_75"], rtol=2e-7, atol=1e-6)
    np.testing.assert_allclose(O.rope_angles(8).numpy(), k["rope_8_torch"], rtol=0, atol=4e-6)


@pytest.mark.parametrize("case", ["small", "cfgrows", "bench1"])
def test_model_case(case, dit_weights):
    g = golden(f"case_{case}.npz")
    w = dit_weights
    ref, ref_len = torch.from_numpy(g["ref"]), torch.from_numpy(g["ref_len"])
    ids, ph_mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["ph_mask"])
    mask, t, x_t = torch.from_numpy(g["mask"]), torch.from_numpy(g["t"]), torch.from_numpy(g["x_t"])
    with torch.no_grad():
        cache = O.encode_conditions(w, ref, ref_len, ids, ph_mask)
        assert np.array_equal(cache["ref_mask"].numpy(), g["ref_mask"])
        assert rel_l2(cache["ref_seq"].numpy(), g["ref_seq"]) < TOL
        ph = O.text_encoder(w, ids, ph_mask)
        valid = g["ph_mask"]
        assert rel_l2(ph.numpy()[valid], g["phoneme_emb"][valid]) < TOL
        for key in [k for k in g if k.startswith("L")]:
            li, name = key[1:].split("_", 1)
            got = cache[name][int(li)].numpy()
            # masked key positions hold don't-care values in both implementations
            km = g["ref_mask"] if name.endswith("ref") else g["ph_mask"]
            sel = np.broadcast_to(km[:, None, :, None], got.shape)
            assert rel_l2(got[sel], g[key][sel]) < TOL, key
        assert rel_l2(O.time_embedding(w, t).numpy(), g["time_emb"]) < TOL
        assert rel_l2(O.input_embedding(w, x_t, mask).numpy(), g["x_embed"]) < TOL
        v = O.denoise_step(w, x_t, mask, t, cache, ph_mask=ph_mask).numpy()
    m = g["mask"]
    assert rel_l2(v[m], g["velocity"][m]) < TOL
    assert rel_l2(v[m], g["velocity_full"][m]) < TOL


def test_sampler4(dit_weights):
    g = golden("case_sampler4.npz")
    w = dit_weights
    B, N = g["noise"].shape[1:3]
    ref, ids = torch.from_numpy(g["ref"]), torch.from_numpy(g["ids"])
    ph_mask = torch.ones_like(ids, dtype=torch.bool)
    mask = torch.ones(B, N, dtype=torch.bool)
    keep = []
    with torch.no_grad():
        cache = O.encode_conditions(w, ref, torch.tensor([ref.shape[1]]), ids, ph_mask)
        O.sample_dmd(w, cache, ph_mask, mask, torch.from_numpy(g["noise"]), 4, keep)
    for i, x in enumerate(keep):
        assert rel_l2(x.numpy(), g["x_pred_steps"][i]) < 5e-5, i


def test_symbol_table():
    from smalltts_amd.phonemes import p2idx, phoneme_len, symbols
    with open(os.path.join(GOLDEN, "symbol_table.json")) as f:
        ref = json.load(f)
    assert phoneme_len == ref["phoneme_len"] == 198
    assert symbols == ref["symbols"]
    # SURVEY §8c known answers
    for sym, idx in {";": 1, " ": 14, "a": 41, "[babble]": 175, "[laughter]": 188, "[whistle]": 197}.items():
        assert p2idx[sym] == idx
