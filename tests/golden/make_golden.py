"""Generate golden vectors from the REFERENCE's own PyTorch modules (build container only).

    python tests/golden/make_golden.py

Imports `/root/reference/src/smalltts/...` (with the decorative deps stubbed, see
_ref_import.py), loads the seeded synthetic weights of `smalltts_amd.weights` into the
reference `DiTModel`, runs the reference's `encode_conditions`, `denoise_step`, `forward`,
`_get_alpha_sigma`, `_compute_rope_freqs`, `get_alpha_sigma` on seeded inputs and stores
inputs + outputs as small .npz fixtures next to this file. Nothing here is needed (or
present) on the GPU box; the fixtures are data only.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402
from smalltts_amd.weights import dit_param_specs, synth_state_dict  # noqa: E402

SEED = 20260928
torch.manual_seed(0)
torch.set_num_threads(8)


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen)


def main():
    DiTModel, ref_infer, ref_tu, ref_ph = import_reference()
    specs = dit_param_specs()
    sd_np = synth_state_dict(specs, SEED)
    model = DiTModel(64).eval()
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == [n for n, _ in specs]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})

    # ---- inventory fixture (names + shapes of the reference state_dict) ----------------
    with open(os.path.join(HERE, "state_dict_inventory.json"), "w") as f:
        json.dump({"n_tensors": len(ref_sd), "n_params": int(sum(v.numel() for v in ref_sd.values())),
                   "tensors": [[k, list(v.shape)] for k, v in ref_sd.items()]}, f)

    # ---- KATs: schedule, rope, symbol table -------------------------------------------
    ts4 = np.linspace(1, 0, 4, dtype=np.float32)
    kat = {
        "ts4": ts4,
        "alpha_sigma_4": np.array([ref_infer._get_alpha_sigma(float(t)) for t in ts4], dtype=np.float32),
        "ts_dense": np.linspace(1, 0, 128, dtype=np.float32),
        "rope_8": ref_infer._compute_rope_freqs(8),
        "rope_75": ref_infer._compute_rope_freqs(75),
    }
    kat["alpha_sigma_dense"] = np.array(
        [ref_infer._get_alpha_sigma(float(t)) for t in kat["ts_dense"]], dtype=np.float32)
    a32, s32 = ref_tu.get_alpha_sigma(torch.from_numpy(ts4))
    kat["alpha_sigma_4_torch_f32"] = np.stack([a32.numpy(), s32.numpy()], -1)
    # torch rotary table of the DiT (dit.py:138-149) for the first 8 positions
    kat["rope_8_torch"] = model.dit.rotary_embed.freqs[:, :8].numpy()
    np.savez(os.path.join(HERE, "kat_schedule_rope.npz"), **kat)
    with open(os.path.join(HERE, "symbol_table.json"), "w") as f:
        json.dump({"phoneme_len": ref_ph.phoneme_len, "symbols": ref_ph.phonemes,
                   "nv_repeat": ref_ph.NV_REPEAT}, f, ensure_ascii=False)

    # ---- model cases -------------------------------------------------------------------
    def run_case(name, B, N, R, P, ref_len, ph_valid, n_valid, t, ids=None, keep_layers=(0, 11), seed=1):
        g = torch.Generator().manual_seed(seed)
        x_t = rnd(g, B, N, 64)
        ref = rnd(g, B, R, 64)
        if ids is None:
            ids = torch.randint(1, 198, (B, P), generator=g)
        ref_len = torch.tensor(ref_len, dtype=torch.int64)
        ph_mask = torch.arange(P)[None, :] < torch.tensor(ph_valid)[:, None]
        ids = ids * ph_mask  # padded ids are 0, as in distill.py:88
        mask = torch.arange(N)[None, :] < torch.tensor(n_valid)[:, None]
        t = torch.tensor(t, dtype=torch.float32)
        with torch.no_grad():
            cache = model.encode_conditions(ref, ref_len, ids, ph_mask, N)
            vel_cached = model.denoise_step(x_t, mask, t, cache)
            vel_full = model(x_t, ref, ref_len, mask, ids, ph_mask, t)
            ref_seq, ref_mask = model.style_encoder(ref, ref_len)
            ph_emb = model.phoneme_embedding(ids, ph_mask)
            temb = model.time_embedding(t)
            x0 = model.dit.input_embed(x_t, mask)
        out = dict(x_t=x_t, ref=ref, ref_len=ref_len, ids=ids, ph_mask=ph_mask, mask=mask, t=t,
                   velocity=vel_cached, velocity_full=vel_full, ref_seq=torch.nan_to_num(ref_seq),
                   ref_mask=ref_mask, phoneme_emb=torch.nan_to_num(ph_emb), time_emb=temb, x_embed=x0)
        for li in keep_layers:
            lay = cache["layers"][li]
            for k in ("k_ref", "v_ref", "k_text", "v_text"):
                out[f"L{li}_{k}"] = lay[k]
        # cheap whole-cache pin: per-layer mean |.| over valid positions
        stats = []
        for lay in cache["layers"]:
            stats.append([float(lay[k].abs().mean()) for k in ("k_ref", "v_ref", "k_text", "v_text")])
        out["cache_absmean"] = torch.tensor(stats)
        np.savez(os.path.join(HERE, f"case_{name}.npz"), **{k: v.numpy() for k, v in out.items()})
        d = float((vel_cached - vel_full).abs().max())
        print(f"case {name}: |cached-full|max = {d:.3e}  |v| rms = {float(vel_cached.pow(2).mean().sqrt()):.4f}")
        return out

    run_case("small", B=2, N=12, R=5, P=7, ref_len=[5, 3], ph_valid=[7, 4], n_valid=[12, 9], t=[0.7, 0.3])
    # CFG-shaped rows: row1 = text dropped (no valid phoneme), row2 = speaker dropped (len 0)
    run_case("cfgrows", B=3, N=20, R=9, P=11, ref_len=[9, 9, 0], ph_valid=[11, 0, 11],
             n_valid=[20, 20, 17], t=[0.5, 0.5, 0.5], seed=2)
    run_case("bench1", B=1, N=75, R=15, P=30, ref_len=[15], ph_valid=[30], n_valid=[75],
             t=[2.0 / 3.0], ids=torch.arange(1, 31)[None], keep_layers=(0,), seed=3)

    # ---- S1: the 4-step re-noising loop of SmallTTS.synthesize (infer/onnx.py:98-125), with
    # the two ONNX calls replaced by the PyTorch modules they were exported from and the
    # noise injected instead of np.random.randn --------------------------------------------
    g = torch.Generator().manual_seed(7)
    B, N, R, P = 1, 12, 5, 7
    ref = rnd(g, B, R, 64)
    ids = torch.randint(1, 198, (B, P), generator=g)
    noise = rnd(g, 4, B, N, 64)
    ref_len = torch.tensor([R])
    ph_mask = torch.ones(B, P, dtype=torch.bool)
    mask = torch.ones(B, N, dtype=torch.bool)
    xs = []
    with torch.no_grad():
        cache = model.encode_conditions(ref, ref_len, ids, ph_mask, N)
        x_pred = np.zeros((B, N, 64), dtype=np.float32)
        for i, t_val in enumerate(np.linspace(1, 0, ref_infer.NUM_STEPS, dtype=np.float32)):
            alpha, sigma = ref_infer._get_alpha_sigma(float(t_val))
            x_t = (alpha * x_pred + sigma * noise[i].numpy()).astype(np.float32)
            v = model.denoise_step(torch.from_numpy(x_t), mask, torch.tensor([t_val]), cache).numpy()
            x_pred = (alpha * x_t - sigma * v).astype(np.float32)
            xs.append(x_pred.copy())
    np.savez(os.path.join(HERE, "case_sampler4.npz"), ref=ref.numpy(), ids=ids.numpy(), noise=noise.numpy(),
             x_pred_steps=np.stack(xs))
    print("sampler4: final rms", float(np.sqrt((xs[-1] ** 2).mean())))
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump({"weights_seed": SEED, "torch": torch.__version__, "numpy": np.__version__}, f)


if __name__ == "__main__":
    main()
