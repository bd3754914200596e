"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the codec
(latents <-> waveform) as specified by `smalltts_amd.weights.CodecSpec`.

Parity status: PARITY UNPINNED against the reference.  The reference has no codec source and no
weights in-tree — only the two ONNX call sites (src/smalltts/codec/onnx.py:34-75) and the
attribution to microsoft/VibeVoice (codec/onnx.py:1); the .onnx files are un-versioned HuggingFace
downloads (src/smalltts/assets/ensure.py:21-40) executed by onnxruntime 1.22.1 (uv.lock).  The
architecture below restates the published VibeVoice acoustic tokenizer design (causal
ConvNeXt-style blocks: RMSNorm -> depthwise causal conv k=7 -> layer-scale residual; RMSNorm ->
Linear 4x -> GELU -> Linear -> layer-scale residual; ConvTranspose1d(k=2r, stride=r) upsampling,
ratios 8,5,5,4,2,2 => hop 3200 = reference HOP_SIZE, infer/onnx.py:12).  What is pinned by the
reference: the I/O contract (B,T,64) <-> (B,1,3200*T).  Written with torch conv ops, i.e. a
different formulation from the GEMM-over-overlapping-rows product kernels.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from smalltts_amd.weights import CodecSpec

W = Dict[str, torch.Tensor]


def _rms_c(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm over the channel axis of (B, C, T)."""
    return x * torch.rsqrt(x.pow(2).mean(1, keepdim=True) + eps) * w[None, :, None]


def _opt(w: W, name: str):
    """Optional tensor (CodecSpec flags conv_bias / ffn_bias / layer_scale / final_norm): None when absent."""
    return w.get(name)


def _scale(w: W, name: str, y: torch.Tensor) -> torch.Tensor:
    g = _opt(w, name)
    return y if g is None else g[None, :, None] * y


def _causal_conv(x, w, b, stride=1, groups=1):
    k = w.shape[-1]
    return F.conv1d(F.pad(x, (k - stride, 0)), w, b, stride=stride, groups=groups)


def _block(w: W, p: str, x: torch.Tensor, spec: CodecSpec) -> torch.Tensor:
    c = x.shape[1]
    n = _rms_c(x, w[f"{p}.norm.weight"], spec.eps)
    y = _causal_conv(n, w[f"{p}.mixer.weight"][:, None, :], _opt(w, f"{p}.mixer.bias"), groups=c)
    x = x + _scale(w, f"{p}.gamma", y)
    n = _rms_c(x, w[f"{p}.ffn_norm.weight"], spec.eps).transpose(1, 2)
    b1, b2 = _opt(w, f"{p}.ffn.w1.bias"), _opt(w, f"{p}.ffn.w2.bias")
    h = n @ w[f"{p}.ffn.w1.weight"].t()
    h = F.gelu(h if b1 is None else h + b1)
    y = h @ w[f"{p}.ffn.w2.weight"].t()
    y = (y if b2 is None else y + b2).transpose(1, 2)
    return x + _scale(w, f"{p}.ffn_gamma", y)


def decode(w: W, latents: torch.Tensor, spec: CodecSpec) -> torch.Tensor:
    """(B, T, latent) -> (B, 1, hop*T)"""
    x = latents.transpose(1, 2)
    x = _causal_conv(x, w["codec.decoder.stem.weight"], _opt(w, "codec.decoder.stem.bias"))
    for i in range(spec.n_stages):
        if i > 0:
            r = spec.ratios[i - 1]
            t_in = x.shape[-1]
            y = F.conv_transpose1d(x, w[f"codec.decoder.up.{i}.weight"], _opt(w, f"codec.decoder.up.{i}.bias"), stride=r)
            x = y[..., : t_in * r]  # causal: trim the (k - stride) trailing samples
        for j in range(spec.dec_depths[i]):
            x = _block(w, f"codec.decoder.stages.{i}.{j}", x, spec)
    if _opt(w, "codec.decoder.final_norm.weight") is not None:
        x = _rms_c(x, w["codec.decoder.final_norm.weight"], spec.eps)
    return _causal_conv(x, w["codec.decoder.head.weight"], _opt(w, "codec.decoder.head.bias"))


def encode(w: W, audio: torch.Tensor, spec: CodecSpec) -> torch.Tensor:
    """(B, 1, S) -> (B, S // hop, latent)"""
    s_use = (audio.shape[-1] // spec.hop) * spec.hop
    x = _causal_conv(audio[..., :s_use], w["codec.encoder.stem.weight"], _opt(w, "codec.encoder.stem.bias"))
    for i in range(spec.n_stages):
        if i > 0:
            r = spec.enc_ratios[i - 1]
            x = _causal_conv(x, w[f"codec.encoder.down.{i}.weight"], _opt(w, f"codec.encoder.down.{i}.bias"), stride=r)
        for j in range(spec.enc_depths[i]):
            x = _block(w, f"codec.encoder.stages.{i}.{j}", x, spec)
    if _opt(w, "codec.encoder.final_norm.weight") is not None:
        x = _rms_c(x, w["codec.encoder.final_norm.weight"], spec.eps)
    x = _causal_conv(x, w["codec.encoder.head.weight"], _opt(w, "codec.encoder.head.bias"))
    return x.transpose(1, 2)
