"""Weight contract for the smalltts hot path: parameter inventory, the seeded
synthetic-weight recipe, and the flat weight-file format.

The reference ships no weights in-tree (they are HuggingFace downloads,
reference `src/smalltts/assets/ensure.py:21-40`), so everything here works from
the *shape contract* of `DiTModel(64).state_dict()` (reference
`src/smalltts/models/backbone/model.py:33-54`, `dit.py:256-285`,
`style.py:108-141`, `phonemes.py:170-198`): 592 tensors, 327,756,609 params.

Synthetic recipe (used by tests, bench and smoke; reproducible bit-for-bit in
numpy here and on the GPU by `smtts_synth_tensor`):

    key   = fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15 mod 2^64)
    z_i   = splitmix64(key + i * 0x9E3779B97F4A7C15)           i = flat index
    s_i   = float32(z_i >> 40) * 2^-23 - 1                     in [-1, 1)
    w_i   = mean + half_range * s_i                            (mul, then add, fp32)

Init scales override the reference's degenerate zero-init of the AdaLN linears
and the velocity head (`dit.py:281-285`, `model.py:53-54`), which would make
velocity == 0 and every parity test vacuous.
"""
from __future__ import annotations

import json
import math
import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15

# ----------------------------------------------------------------------------
# model hyper-parameters (reference model.py:36-50, style.py:108-115)
# ----------------------------------------------------------------------------
LATENT_DIM = 64
HIDDEN = 960
N_BLOCKS = 12
N_HEADS = 8
HEAD_DIM = 120
FF_HIDDEN = 2400
ROPE_DIM = 64
CONV_K = 31
CONV_GROUPS = 16
PHONEME_VOCAB = 198  # reference data/phonemization/phonemes.py:54
TEXT_DIM = 512
TEXT_LAYERS = 8
TEXT_HEADS = 4
TEXT_FF = 1024
TEXT_EPS = 1e-6
STYLE_DIM = 512
STYLE_LAYERS = 12
STYLE_HEADS = 8
STYLE_FF = 1536
STYLE_EPS = 1e-5
TIME_SIN_DIM = 256


def _encoder_block_specs(prefix: str, dim: int, heads: int, ff: int):
    dh = dim // heads
    out = []
    for nm in ("wq", "wk", "wv", "wo", "gate"):
        out.append((f"{prefix}.attention.{nm}.weight", (dim, dim)))
    out.append((f"{prefix}.attention.q_norm.weight", (heads, dh)))
    out.append((f"{prefix}.attention.k_norm.weight", (heads, dh)))
    out.append((f"{prefix}.mlp.w1.weight", (ff, dim)))
    out.append((f"{prefix}.mlp.w3.weight", (ff, dim)))
    out.append((f"{prefix}.mlp.w2.weight", (dim, ff)))
    out.append((f"{prefix}.attention_norm.weight", (dim,)))
    out.append((f"{prefix}.mlp_norm.weight", (dim,)))
    return out


def dit_param_specs() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) for every tensor of the reference `DiTModel(64).state_dict()`,
    in state_dict order (module registration order of model.py:39-51)."""
    s: List[Tuple[str, Tuple[int, ...]]] = []
    # TimeEmbedding (model.py:16-20)
    s += [("time_embedding.mlp.0.weight", (HIDDEN, TIME_SIN_DIM)),
          ("time_embedding.mlp.0.bias", (HIDDEN,)),
          ("time_embedding.mlp.2.weight", (HIDDEN, HIDDEN)),
          ("time_embedding.mlp.2.bias", (HIDDEN,))]
    # TextEncoder (phonemes.py:170-198)
    s.append(("phoneme_embedding.text_embedding.weight", (PHONEME_VOCAB, TEXT_DIM)))
    for i in range(TEXT_LAYERS):
        s += _encoder_block_specs(f"phoneme_embedding.blocks.{i}", TEXT_DIM, TEXT_HEADS, TEXT_FF)
    s.append(("phoneme_embedding.norm.weight", (TEXT_DIM,)))
    # StyleEncoder (style.py:119-141): log_scale is registered after blocks? No:
    # in_proj, blocks, log_scale, norm, out_proj  -- but nn.Module.state_dict lists
    # *parameters of a module before its children*, so log_scale comes first.
    s.append(("style_encoder.log_scale", ()))
    s += [("style_encoder.in_proj.weight", (STYLE_DIM, LATENT_DIM)),
          ("style_encoder.in_proj.bias", (STYLE_DIM,))]
    for i in range(STYLE_LAYERS):
        s += _encoder_block_specs(f"style_encoder.blocks.{i}", STYLE_DIM, STYLE_HEADS, STYLE_FF)
    s += [("style_encoder.norm.weight", (STYLE_DIM,)),
          ("style_encoder.out_proj.weight", (HIDDEN, STYLE_DIM)),
          ("style_encoder.out_proj.bias", (HIDDEN,))]
    # DiT (dit.py:256-285)
    s += [("dit.input_embed.proj.weight", (HIDDEN, LATENT_DIM)),
          ("dit.input_embed.proj.bias", (HIDDEN,)),
          ("dit.input_embed.conv_pos_embed.conv1.weight", (HIDDEN, HIDDEN // CONV_GROUPS, CONV_K)),
          ("dit.input_embed.conv_pos_embed.conv1.bias", (HIDDEN,)),
          ("dit.input_embed.conv_pos_embed.conv2.weight", (HIDDEN, HIDDEN // CONV_GROUPS, CONV_K)),
          ("dit.input_embed.conv_pos_embed.conv2.bias", (HIDDEN,)),
          ("dit.phoneme_proj.weight", (HIDDEN, TEXT_DIM)),
          ("dit.phoneme_proj.bias", (HIDDEN,)),
          ("dit.emb_proj.0.weight", (2 * HIDDEN, HIDDEN)),
          ("dit.emb_proj.0.bias", (2 * HIDDEN,)),
          ("dit.emb_proj.2.weight", (HIDDEN, 2 * HIDDEN)),
          ("dit.emb_proj.2.bias", (HIDDEN,))]
    for i in range(N_BLOCKS):
        p = f"dit.transformer_blocks.{i}"
        s += [(f"{p}.attn_norm.linear.weight", (6 * HIDDEN, HIDDEN)),
              (f"{p}.attn_norm.linear.bias", (6 * HIDDEN,))]
        for nm in ("to_q", "to_k_self", "to_v_self"):
            s += [(f"{p}.attn.{nm}.weight", (HIDDEN, HIDDEN)), (f"{p}.attn.{nm}.bias", (HIDDEN,))]
        s.append((f"{p}.attn.gate.weight", (HIDDEN, HIDDEN)))
        s.append((f"{p}.attn.to_out.0.weight", (HIDDEN, HIDDEN)))
        s.append((f"{p}.attn.q_norm.weight", (N_HEADS, HEAD_DIM)))
        s.append((f"{p}.attn.k_norm.weight", (N_HEADS, HEAD_DIM)))
        for nm in ("to_k_ref", "to_v_ref", "to_k_text", "to_v_text"):
            s += [(f"{p}.attn.{nm}.weight", (HIDDEN, HIDDEN)), (f"{p}.attn.{nm}.bias", (HIDDEN,))]
        s.append((f"{p}.attn.k_norm_cross.weight", (N_HEADS, HEAD_DIM)))
        s += [(f"{p}.ff.w1.weight", (FF_HIDDEN, HIDDEN)), (f"{p}.ff.w1.bias", (FF_HIDDEN,)),
              (f"{p}.ff.w3.weight", (FF_HIDDEN, HIDDEN)), (f"{p}.ff.w3.bias", (FF_HIDDEN,)),
              (f"{p}.ff.w2.weight", (HIDDEN, FF_HIDDEN)), (f"{p}.ff.w2.bias", (HIDDEN,))]
    s += [("dit.norm_out.linear.weight", (2 * HIDDEN, HIDDEN)),
          ("dit.norm_out.linear.bias", (2 * HIDDEN,)),
          ("velocity.weight", (LATENT_DIM, HIDDEN)),
          ("velocity.bias", (LATENT_DIM,))]
    return s


# ----------------------------------------------------------------------------
# codec spec (build-defined: the reference has no codec source, only the ONNX
# call sites `codec/onnx.py:34-75`; SURVEY §7 hard part 1)
# ----------------------------------------------------------------------------
class CodecSpec:
    """Causal conv/ConvNeXt-style latent<->waveform codec in the shape of the
    VibeVoice acoustic tokenizer the reference's ONNX files were exported from
    (`codec/onnx.py:1`). Every hyper-parameter is a field so real weights can be
    dropped in later. Decoder: stem conv(k) latent->C0; per stage i: [upsample
    ConvTranspose1d(k=2r, stride=r) for i>0], depth[i] blocks; head conv(k)->1.
    hop = prod(ratios) must equal the reference HOP_SIZE (infer/onnx.py:12)."""

    def __init__(self, latent_dim=64, n_filters=32, ratios=(8, 5, 5, 4, 2, 2),
                 dec_depths=(8, 3, 3, 3, 3, 3, 3), kernel=7, ffn_mult=4, eps=1e-5,
                 conv_bias=True, ffn_bias=True, layer_scale=True, final_norm=False):
        self.latent_dim = latent_dim
        self.n_filters = n_filters
        self.ratios = tuple(ratios)          # decoder order (coarse -> fine)
        self.dec_depths = tuple(dec_depths)  # decoder order
        self.kernel = kernel
        self.ffn_mult = ffn_mult
        self.eps = eps
        # Which OPTIONAL tensors the parameter inventory contains.  The engine and the oracle treat them as optional at run
        # time (absent bias = 0, absent layer scale = 1, absent final norm = identity), so an exported codec that differs from
        # the default in these respects loads without code changes; the flags only steer the inventory (synthetic weights,
        # converter report).
        self.conv_bias = bool(conv_bias)      # bias of stem / head / resampling / depthwise convs
        self.ffn_bias = bool(ffn_bias)        # bias of the two FFN linears
        self.layer_scale = bool(layer_scale)  # per-channel gamma / ffn_gamma on the two residual branches
        self.final_norm = bool(final_norm)    # RMSNorm in front of the head conv
        assert len(self.dec_depths) == len(self.ratios) + 1

    @property
    def hop(self) -> int:
        return int(np.prod(self.ratios))

    @property
    def n_stages(self) -> int:
        return len(self.dec_depths)

    def dec_channels(self, stage: int) -> int:
        return self.n_filters * (2 ** (self.n_stages - 1 - stage))

    def to_dict(self):
        return dict(latent_dim=self.latent_dim, n_filters=self.n_filters, ratios=list(self.ratios),
                    dec_depths=list(self.dec_depths), kernel=self.kernel, ffn_mult=self.ffn_mult,
                    eps=self.eps, conv_bias=self.conv_bias, ffn_bias=self.ffn_bias, layer_scale=self.layer_scale,
                    final_norm=self.final_norm)

    # encoder mirrors the decoder: stage order fine -> coarse
    @property
    def enc_ratios(self):
        return tuple(reversed(self.ratios))

    @property
    def enc_depths(self):
        return tuple(reversed(self.dec_depths))

    def enc_channels(self, stage: int) -> int:
        return self.n_filters * (2 ** stage)


DEFAULT_CODEC = CodecSpec()


def _optional(spec: CodecSpec, name: str) -> bool:
    """Is this tensor of the full inventory present under the spec's flags?"""
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("gamma", "ffn_gamma"):
        return spec.layer_scale
    if leaf == "bias":
        return spec.ffn_bias if ".ffn." in name else spec.conv_bias
    if ".final_norm." in name:
        return spec.final_norm
    return True


def _block_specs(prefix: str, c: int, spec: CodecSpec):
    k, f = spec.kernel, spec.ffn_mult
    full = [(f"{prefix}.norm.weight", (c,)),
            (f"{prefix}.mixer.weight", (c, k)),      # depthwise causal conv
            (f"{prefix}.mixer.bias", (c,)),
            (f"{prefix}.gamma", (c,)),
            (f"{prefix}.ffn_norm.weight", (c,)),
            (f"{prefix}.ffn.w1.weight", (f * c, c)),
            (f"{prefix}.ffn.w1.bias", (f * c,)),
            (f"{prefix}.ffn.w2.weight", (c, f * c)),
            (f"{prefix}.ffn.w2.bias", (c,)),
            (f"{prefix}.ffn_gamma", (c,))]
    return [t for t in full if _optional(spec, t[0])]


def codec_decoder_param_specs(spec: CodecSpec = DEFAULT_CODEC):
    s = []
    c0 = spec.dec_channels(0)
    s += [("codec.decoder.stem.weight", (c0, spec.latent_dim, spec.kernel)),
          ("codec.decoder.stem.bias", (c0,))]
    for i in range(spec.n_stages):
        c = spec.dec_channels(i)
        if i > 0:
            r = spec.ratios[i - 1]
            # ConvTranspose1d weight layout (Cin, Cout, K) like torch
            s += [(f"codec.decoder.up.{i}.weight", (2 * c, c, 2 * r)),
                  (f"codec.decoder.up.{i}.bias", (c,))]
        for j in range(spec.dec_depths[i]):
            s += _block_specs(f"codec.decoder.stages.{i}.{j}", c, spec)
    cl = spec.dec_channels(spec.n_stages - 1)
    s += [("codec.decoder.final_norm.weight", (cl,)),
          ("codec.decoder.head.weight", (1, cl, spec.kernel)), ("codec.decoder.head.bias", (1,))]
    return [t for t in s if _optional(spec, t[0])]


def codec_encoder_param_specs(spec: CodecSpec = DEFAULT_CODEC):
    s = []
    c0 = spec.enc_channels(0)
    s += [("codec.encoder.stem.weight", (c0, 1, spec.kernel)), ("codec.encoder.stem.bias", (c0,))]
    for i in range(spec.n_stages):
        c = spec.enc_channels(i)
        if i > 0:
            r = spec.enc_ratios[i - 1]
            s += [(f"codec.encoder.down.{i}.weight", (c, c // 2, 2 * r)),
                  (f"codec.encoder.down.{i}.bias", (c,))]
        for j in range(spec.enc_depths[i]):
            s += _block_specs(f"codec.encoder.stages.{i}.{j}", c, spec)
    cl = spec.enc_channels(spec.n_stages - 1)
    s += [("codec.encoder.final_norm.weight", (cl,)),
          ("codec.encoder.head.weight", (spec.latent_dim, cl, spec.kernel)),
          ("codec.encoder.head.bias", (spec.latent_dim,))]
    return [t for t in s if _optional(spec, t[0])]


def all_param_specs(spec: CodecSpec = DEFAULT_CODEC):
    return dit_param_specs() + codec_decoder_param_specs(spec) + codec_encoder_param_specs(spec)


# ----------------------------------------------------------------------------
# synthetic recipe
# ----------------------------------------------------------------------------
def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & MASK64
    return h


def tensor_key(name: str, seed: int) -> int:
    return fnv1a64(name) ^ ((seed * GOLDEN) & MASK64)


def init_rule(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, half_range) of the uniform synthetic init for a tensor."""
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 0:                                   # style_encoder.log_scale (style.py:134)
        return -1.8, 0.1
    if leaf in ("gamma", "ffn_gamma"):                    # codec layer scales
        return 0.5, 0.2
    is_norm = (name.endswith("norm.weight") or name.endswith("norm_cross.weight"))
    if is_norm:
        return 1.0, 0.2
    if leaf == "bias":
        return 0.0, 0.1
    if name.endswith("text_embedding.weight"):
        return 0.0, math.sqrt(3.0)
    if ".up." in name and leaf == "weight":
        # ConvTranspose (Cin, Cout, K): each output sample sums Cin * (K/stride = 2) taps
        fan_in = shape[0] * 2
        return 0.0, math.sqrt(3.0 / fan_in)
    fan_in = int(np.prod(shape[1:]))
    return 0.0, math.sqrt(3.0 / fan_in)


def synth_uniform(key: int, n: int) -> np.ndarray:
    """s_i in [-1,1) as float32, i = 0..n-1 (see module docstring)."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(GOLDEN) + np.uint64(key & MASK64)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    u = (z >> np.uint64(40)).astype(np.float32)
    return u * np.float32(2.0 ** -23) - np.float32(1.0)


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    mean, hr = init_rule(name, shape)
    n = int(np.prod(shape)) if len(shape) else 1
    s = synth_uniform(tensor_key(name, seed), n)
    w = np.float32(mean) + np.float32(hr) * s
    return w.astype(np.float32).reshape(shape)


def synth_state_dict(specs: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, np.ndarray]:
    return {name: synth_tensor(name, shape, seed) for name, shape in specs}


# ----------------------------------------------------------------------------
# checkpoint key clean-up (same prefixes the reference strips, distill.py:47-54)
# ----------------------------------------------------------------------------
_PREFIXES = ("module.", "_orig_mod.", "ema_model.", "online_model.")


def clean_state_dict_keys(sd: Dict[str, object]) -> Dict[str, object]:
    out = {}
    for k, v in sd.items():
        if k in ("initted", "step"):
            continue
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k = k[len(p):]
                    changed = True
        out[k.replace("._orig_mod.", ".")] = v
    return out


# ----------------------------------------------------------------------------
# flat weight file:  b"SMTTSW01" | u64 json_len | json | pad to 64 | raw fp32 LE
# json = {"tensors": [{"name","shape","offset"}...], "codec": {...}|null}
# ----------------------------------------------------------------------------
MAGIC = b"SMTTSW01"


def save_weight_file(path: str, tensors: Dict[str, np.ndarray], codec: CodecSpec | None = None) -> None:
    table, off = [], 0
    for name, arr in tensors.items():
        n = int(arr.size)
        table.append({"name": name, "shape": [int(d) for d in arr.shape], "offset": off})
        off += ((n * 4 + 63) // 64) * 64
    hdr = json.dumps({"tensors": table, "codec": codec.to_dict() if codec else None}).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(hdr)))
        f.write(hdr)
        pos = 16 + len(hdr)
        f.write(b"\0" * ((-pos) % 64))
        for name, arr in tensors.items():
            b = np.ascontiguousarray(arr, dtype="<f4").tobytes()
            f.write(b)
            f.write(b"\0" * ((-len(b)) % 64))


def load_weight_file(path: str) -> Tuple[Dict[str, np.ndarray], dict | None]:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a smalltts weight file")
        (n,) = struct.unpack("<Q", f.read(8))
        meta = json.loads(f.read(n))
        base = 16 + n
        base += (-base) % 64
        out = {}
        for t in meta["tensors"]:
            cnt = int(np.prod(t["shape"])) if t["shape"] else 1
            f.seek(base + t["offset"])
            out[t["name"]] = np.frombuffer(f.read(cnt * 4), dtype="<f4").reshape(t["shape"]).copy()
    return out, meta.get("codec")
