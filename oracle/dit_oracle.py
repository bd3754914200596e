"""ORACLE (test infrastructure, NOT product code).

CPU fp32 restatement of the reference's hot path: condition encoders, the cached
DiT denoiser, the DMD re-noising sampler and the build-defined teacher ODE sampler.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module; `smalltts_amd/` never does.

Parity status: PINNED. `tests/golden/make_golden.py` imports the reference's own
PyTorch modules (`/root/reference/src/smalltts/models/backbone/*.py`) in the build
container, loads the seeded synthetic weights of `smalltts_amd/weights.py` into
them and records inputs/outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this file against those vectors. The teacher
ODE sampler (S2) has no reference loop and is build-defined; its CFG
combination and x0-prediction follow `src/scripts/train/dmd2/distill.py:60-134`.

Written functionally over a flat {name: tensor} dict with the reference's
state_dict names; it is not a copy of the reference's nn.Module code.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]

HIDDEN, HEADS, DH, NBLK, ROPE_DIM = 960, 8, 120, 12, 64


# --- S1a: schedule (reference infer/onnx.py:31-39; float64 math, cast to f32) ---------
def alpha_sigma(t: float, eps: float = 1e-5) -> Tuple[np.float32, np.float32]:
    t = min(max(float(t), eps), 1.0 - eps)
    a2 = math.cos(math.pi / 2.0 * t) ** 2
    lsnr = math.log(a2 / (1.0 - a2)) + 2.0 * math.log(0.5)
    asq = 1.0 / (1.0 + math.exp(-lsnr))
    return np.float32(math.sqrt(asq)), np.float32(math.sqrt(1.0 - asq))


# --- S1b: rope angle table (reference infer/onnx.py:42-47 == dit.py:138-149) ----------
def rope_angles(n: int, dim: int = ROPE_DIM) -> torch.Tensor:
    inv = 1.0 / (1e4 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    ang = torch.arange(n, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.repeat_interleave(2, dim=-1)[None]  # (1, n, dim): a0 a0 a1 a1 ...


def _rot_pairs(x: torch.Tensor, ang_half: torch.Tensor) -> torch.Tensor:
    """Rotate (x[2i], x[2i+1]) by ang_half[..., i]. Both RoPE flavours of the reference
    reduce to this: dit.py:152-173 (rotate_half on interleaved pairs) and
    style.py:21-25 / phonemes.py:79-83 (complex multiply on (2i,2i+1))."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = ang_half.cos(), ang_half.sin()
    out = torch.empty_like(x)
    out[..., 0::2] = xe * c - xo * s
    out[..., 1::2] = xo * c + xe * s
    return out


def _rms(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # dit.py:42-53 : normalise over the last dim, multiply by weight (1-D or (h, dh))
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _ln(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def _lin(w: W, name: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    y = x @ w[name + ".weight"].t()
    return y + w[name + ".bias"] if bias else y


def _sdpa(q, k, v, key_mask: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh) + mask) v ; q (B,H,Nq,dh) k,v (B,H,Nk,dh) key_mask (B,Nk) bool.
    A row whose keys are all masked yields 0 here; the reference yields NaN or 0 depending on
    the SDPA backend, but every such row is zeroed afterwards (style.py:173, dit.py:295-297)."""
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    m = s.amax(-1, keepdim=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = (s - m).exp()
    den = p.sum(-1, keepdim=True)
    p = torch.where(den > 0, p / den.clamp_min(1e-30), torch.zeros_like(p))
    return p @ v


# --- K10: encoder block (style.py:28-105, phonemes.py:87-167) -------------------------
def _encoder_block(w: W, p: str, x, key_mask, heads: int, eps: float, ang_half):
    b, n, d = x.shape
    dh = d // heads
    h = _rms(x, w[f"{p}.attention_norm.weight"], eps)
    q = _lin(w, f"{p}.attention.wq", h, False).reshape(b, n, heads, dh)
    k = _lin(w, f"{p}.attention.wk", h, False).reshape(b, n, heads, dh)
    v = _lin(w, f"{p}.attention.wv", h, False).reshape(b, n, heads, dh)
    g = _lin(w, f"{p}.attention.gate", h, False)
    q = _rms(q, w[f"{p}.attention.q_norm.weight"], eps)
    k = _rms(k, w[f"{p}.attention.k_norm.weight"], eps)
    q = _rot_pairs(q, ang_half[None, :n, None, :])
    k = _rot_pairs(k, ang_half[None, :n, None, :])
    o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), key_mask)
    o = o.transpose(1, 2).reshape(b, n, d) * torch.sigmoid(g)
    x = x + _lin(w, f"{p}.attention.wo", o, False)
    h = _rms(x, w[f"{p}.mlp_norm.weight"], eps)
    y = F.silu(_lin(w, f"{p}.mlp.w1", h, False)) * _lin(w, f"{p}.mlp.w3", h, False)
    return x + _lin(w, f"{p}.mlp.w2", y, False)


def _enc_angles(n: int, dh: int) -> torch.Tensor:
    inv = 1.0 / (10000.0 ** (torch.arange(0, dh, 2).float() / dh))
    return torch.arange(n).float()[:, None] * inv[None, :]  # (n, dh/2)


# --- E1: style encoder (style.py:144-174) ---------------------------------------------
def style_encoder(w: W, ref: torch.Tensor, ref_len: torch.Tensor):
    b, r, _ = ref.shape
    mask = torch.arange(r)[None, :] < ref_len.clamp(max=r)[:, None]
    x = _lin(w, "style_encoder.in_proj", ref) * w["style_encoder.log_scale"].exp()
    ang = _enc_angles(r, 64)
    for i in range(12):
        x = _encoder_block(w, f"style_encoder.blocks.{i}", x, mask, 8, 1e-5, ang)
    x = _rms(x, w["style_encoder.norm.weight"], 1e-5)
    x = _lin(w, "style_encoder.out_proj", x)
    x = torch.where(mask[..., None], x, torch.zeros_like(x))  # also clears NaN-free zero rows
    return x, mask


# --- E2: text encoder (phonemes.py:200-207) -------------------------------------------
def text_encoder(w: W, ids: torch.Tensor, ph_mask: torch.Tensor):
    x = w["phoneme_embedding.text_embedding.weight"][ids]
    ang = _enc_angles(ids.shape[1], 128)
    for i in range(8):
        x = _encoder_block(w, f"phoneme_embedding.blocks.{i}", x, ph_mask, 4, 1e-6, ang)
    return _rms(x, w["phoneme_embedding.norm.weight"], 1e-6)


# --- E3 / E0: cross-KV cache (dit.py:80-93, 293-314; model.py:88-95) -------------------
def encode_conditions(w: W, ref, ref_len, ids, ph_mask):
    """-> dict(k_ref, v_ref (L,B,H,R,dh), ref_mask (B,R), k_text, v_text (L,B,H,P,dh))."""
    ref_seq, ref_mask = style_encoder(w, ref, ref_len)
    ph = text_encoder(w, ids, ph_mask)
    mem = _lin(w, "dit.phoneme_proj", ph)
    mem = torch.where(ph_mask[..., None], mem, torch.zeros_like(mem))
    out = {k: [] for k in ("k_ref", "v_ref", "k_text", "v_text")}
    for i in range(NBLK):
        p = f"dit.transformer_blocks.{i}.attn"
        for tag, seq in (("ref", ref_seq), ("text", mem)):
            b, n, _ = seq.shape
            k = _lin(w, f"{p}.to_k_{tag}", seq).reshape(b, n, HEADS, DH)
            k = _rms(k, w[f"{p}.k_norm_cross.weight"], 1e-6).transpose(1, 2)
            v = _lin(w, f"{p}.to_v_{tag}", seq).reshape(b, n, HEADS, DH).transpose(1, 2)
            out[f"k_{tag}"].append(k)
            out[f"v_{tag}"].append(v)
    res = {k: torch.stack(v) for k, v in out.items()}
    res["ref_mask"] = ref_mask
    res["ref_seq"] = ref_seq       # intermediates, exposed for layer-wise parity tests
    res["phoneme_mem"] = mem
    return res


# --- D1: time embedding (model.py:16-30) ----------------------------------------------
def time_embedding(w: W, t: torch.Tensor) -> torch.Tensor:
    half = 128
    f = torch.exp(torch.arange(half).float() * -(math.log(1e4) / (half - 1)))
    e = 1e3 * t[:, None] * f[None, :]
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    return _lin(w, "time_embedding.mlp.2", F.silu(_lin(w, "time_embedding.mlp.0", e)))


def _mish(x):
    return x * torch.tanh(F.softplus(x))


# --- D2: input embedding (dit.py:215-253) ---------------------------------------------
def input_embedding(w: W, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    h = _lin(w, "dit.input_embed.proj", x)
    m = mask[..., None].to(h.dtype)
    c = (h * m).transpose(1, 2)
    p = "dit.input_embed.conv_pos_embed"
    c = _mish(F.conv1d(c, w[f"{p}.conv1.weight"], w[f"{p}.conv1.bias"], padding=15, groups=16))
    c = c * m.transpose(1, 2)
    c = _mish(F.conv1d(c, w[f"{p}.conv2.weight"], w[f"{p}.conv2.bias"], padding=15, groups=16))
    return c.transpose(1, 2) * m + h


# --- D0/D3..D9: cached denoiser (model.py:97-100, dit.py:316-327, 189-212, 95-135) -----
def denoise_step(w: W, x_t, mask, t, cache, rope: Optional[torch.Tensor] = None,
                 ph_mask: Optional[torch.Tensor] = None, trace: Optional[dict] = None):
    b, n, _ = x_t.shape
    if rope is None:
        rope = rope_angles(n)
    ang_half = rope[0, :n, 0::2]  # (n, 32)
    temb = time_embedding(w, t)
    x = input_embedding(w, x_t, mask)
    emb = _lin(w, "dit.emb_proj.2", F.silu(_lin(w, "dit.emb_proj.0", temb)))
    semb = F.silu(emb)
    key_mask = torch.cat([mask, cache["ref_mask"], ph_mask], dim=1)
    if trace is not None:
        trace["temb"], trace["emb"], trace["x0"] = temb, emb, x
    for i in range(NBLK):
        p = f"dit.transformer_blocks.{i}"
        mod = _lin(w, f"{p}.attn_norm.linear", semb)
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
        y = _ln(x) * (1 + sc_a[:, None]) + sh_a[:, None]
        q = _lin(w, f"{p}.attn.to_q", y).reshape(b, n, HEADS, DH)
        k = _lin(w, f"{p}.attn.to_k_self", y).reshape(b, n, HEADS, DH)
        v = _lin(w, f"{p}.attn.to_v_self", y).reshape(b, n, HEADS, DH)
        q = _rms(q, w[f"{p}.attn.q_norm.weight"], 1e-6)
        k = _rms(k, w[f"{p}.attn.k_norm.weight"], 1e-6)
        q = torch.cat([_rot_pairs(q[..., :ROPE_DIM], ang_half[None, :, None, :]), q[..., ROPE_DIM:]], -1)
        k = torch.cat([_rot_pairs(k[..., :ROPE_DIM], ang_half[None, :, None, :]), k[..., ROPE_DIM:]], -1)
        kk = torch.cat([k.transpose(1, 2), cache["k_ref"][i], cache["k_text"][i]], dim=2)
        vv = torch.cat([v.transpose(1, 2), cache["v_ref"][i], cache["v_text"][i]], dim=2)
        o = _sdpa(q.transpose(1, 2), kk, vv, key_mask).transpose(1, 2).reshape(b, n, HIDDEN)
        o = o * torch.sigmoid(_lin(w, f"{p}.attn.gate", y, False))
        o = _lin(w, f"{p}.attn.to_out.0", o, False)
        o = torch.where(mask[..., None], o, torch.zeros_like(o))
        x = x + torch.tanh(g_a)[:, None] * o
        nrm = _ln(x) * (1 + sc_m[:, None]) + sh_m[:, None]
        ff = _lin(w, f"{p}.ff.w2", F.silu(_lin(w, f"{p}.ff.w1", nrm)) * _lin(w, f"{p}.ff.w3", nrm))
        x = x + torch.tanh(g_m)[:, None] * ff
        if trace is not None:
            trace[f"x{i + 1}"] = x
    scale, shift = _lin(w, "dit.norm_out.linear", semb).chunk(2, dim=1)  # scale first (dit.py:37)
    x = _ln(x) * (1 + scale[:, None]) + shift[:, None]
    return _lin(w, "velocity", x)


# --- S1: DMD re-noising sampler (infer/onnx.py:98-125) --------------------------------
def sample_dmd(w: W, cache, ph_mask, mask, noise: torch.Tensor, num_steps: int = 4,
               keep: Optional[list] = None) -> torch.Tensor:
    """noise: (num_steps, B, N, 64) injected eps per step (the reference draws it from the
    unseeded global numpy RNG, infer/onnx.py:104)."""
    b, n = mask.shape
    x = torch.zeros(b, n, 64)
    ts = np.linspace(1, 0, num_steps, dtype=np.float32)
    for i, tv in enumerate(ts):
        a, s = alpha_sigma(float(tv))
        x_t = float(a) * x + float(s) * noise[i]
        v = denoise_step(w, x_t, mask, torch.full((b,), float(tv)), cache, ph_mask=ph_mask)
        x = float(a) * x_t - float(s) * v
        if keep is not None:
            keep.append(x.clone())
    return x


# --- S2: teacher sampler (build-defined; CFG per distill.py:60-134) --------------------
def cfg_conditions(ref, ref_len, ids, ph_mask):
    """3B-row condition batch: [cond ; text dropped ; speaker dropped] (distill.py:76-99)."""
    ref3 = torch.cat([ref, ref, torch.zeros_like(ref)], 0)
    len3 = torch.cat([ref_len, ref_len, torch.zeros_like(ref_len)], 0)
    ids3 = torch.cat([ids, torch.zeros_like(ids), ids], 0)
    pm3 = torch.cat([ph_mask, torch.zeros_like(ph_mask), ph_mask], 0)
    return ref3, len3, ids3, pm3


def cfg_velocity(w: W, x_t, mask, t, cache3, pm3, s_text=2.0, s_spk=1.5):
    v3 = denoise_step(w, x_t.repeat(3, 1, 1), mask.repeat(3, 1), t.repeat(3), cache3, ph_mask=pm3)
    vc, vt, vs = v3.chunk(3, 0)
    return vc + s_text * (vc - vt) + s_spk * (vc - vs)  # distill.py:101-105


def sample_teacher_ode(w: W, cache3, pm3, mask, noise0: torch.Tensor, num_steps: int = 128,
                       s_text=2.0, s_spk=1.5, keep: Optional[list] = None) -> torch.Tensor:
    """Deterministic (DDIM-style) ODE over t = linspace(1, 0, num_steps):
    x_1 = sigma(1) * eps; at each t: x0_hat = a x_t - s v (distill.py:127-130),
    eps_hat = s x_t + a v (inverse of train/utils.py:65-66), x_{t'} = a' x0_hat + s' eps_hat."""
    b, n = mask.shape
    ts = np.linspace(1, 0, num_steps, dtype=np.float32)
    a, s = alpha_sigma(float(ts[0]))
    x_t = float(s) * noise0
    x0 = torch.zeros_like(x_t)
    for i, tv in enumerate(ts):
        a, s = alpha_sigma(float(tv))
        v = cfg_velocity(w, x_t, mask, torch.full((b,), float(tv)), cache3, pm3, s_text, s_spk)
        x0 = float(a) * x_t - float(s) * v
        eps = float(s) * x_t + float(a) * v
        if keep is not None:
            keep.append(x0.clone())
        if i + 1 < num_steps:
            a2, s2 = alpha_sigma(float(ts[i + 1]))
            x_t = float(a2) * x0 + float(s2) * eps
    return x0


def to_torch(sd: Dict[str, np.ndarray]) -> W:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
