"""Study (test infrastructure): what operand precision do the GEMM sites of the path need?

Emulates, inside the CPU oracle, GEMMs whose operands are rounded the way a 1- or 2-term MFMA scheme would
round them, and prints the latent rel-L2 (4-step DMD sampler, bench shape) / codec SNR against the exact
fp32 oracle.  Used to choose the per-site precision ladder of round 2 before writing kernels; the GPU tests
are what gate the product.

  python -m tests.studies.precision_ladder_cpu [dit|codec]
"""
import json
import os
import sys

import numpy as np
import torch

from oracle import codec_oracle as CO
from oracle import dit_oracle as O
from smalltts_amd.weights import DEFAULT_CODEC, codec_decoder_param_specs, dit_param_specs, synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def q(x, dt):
    return x.to(dt).to(torch.float32)


def q2(x, dt):
    """two-term split: hi + lo, both in dt"""
    hi = q(x, dt)
    return hi + q(x - hi, dt)


MODES = {
    # name: (A rounding, W rounding)
    "exact": (None, None),
    "bf16x1": (lambda a: q(a, torch.bfloat16), lambda w: q(w, torch.bfloat16)),
    "fp16x1": (lambda a: q(a, torch.float16), lambda w: q(w, torch.float16)),
    "bf16 A2.W1": (lambda a: q2(a, torch.bfloat16), lambda w: q(w, torch.bfloat16)),
    "bf16 A1.W2": (lambda a: q(a, torch.bfloat16), lambda w: q2(w, torch.bfloat16)),
    "fp16 A2.W1": (lambda a: q2(a, torch.float16), lambda w: q(w, torch.float16)),
    "fp16 A1.W2": (lambda a: q(a, torch.float16), lambda w: q2(w, torch.float16)),
    "bf16x3": (lambda a: q2(a, torch.bfloat16), lambda w: q2(w, torch.bfloat16)),
}


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def snr_db(a, b):
    a, b = a.double(), b.double()
    return float(10 * torch.log10((b ** 2).sum() / ((a - b) ** 2).sum()))


def study_dit(seed):
    w = O.to_torch(synth_state_dict(dit_param_specs(), seed))
    B, N, R, P = 8, 75, 15, 30
    g = torch.Generator().manual_seed(1)
    ref = torch.randn(B, R, 64, generator=g)
    ids = torch.arange(1, P + 1)[None].repeat(B, 1)
    rl = torch.full((B,), R)
    pm = torch.ones(B, P, dtype=torch.bool)
    mask = torch.ones(B, N, dtype=torch.bool)
    noise = torch.randn(4, B, N, 64, generator=g)
    orig = O._lin
    site = {"sel": None}

    def run(fa, fw, only=None):
        cache = {}

        def lin(wd, name, x, bias=True):
            if fa is None or (only is not None and not only(name)):
                return orig(wd, name, x, bias)
            if name not in cache:
                cache[name] = fw(wd[name + ".weight"])
            y = fa(x) @ cache[name].t()
            return y + wd[name + ".bias"] if bias else y
        O._lin = lin
        try:
            with torch.no_grad():
                c = O.encode_conditions(w, ref, rl, ids, pm)
                return O.sample_dmd(w, c, pm, mask, noise, 4)
        finally:
            O._lin = orig

    x0 = run(None, None)
    print("DiT latent rel-L2 vs exact fp32 oracle (B=8,N=75,R=15,P=30, 4 DMD steps)")
    for name, (fa, fw) in MODES.items():
        if fa is None:
            continue
        print(f"  all sites {name:12s} {rel_l2(run(fa, fw), x0):.3e}")
    groups = {
        "ff (w1,w3,w2)": lambda n: ".ff.w" in n,
        "qkvg+out": lambda n: ".attn.to_" in n or ".attn.gate" in n,
        "encoders": lambda n: n.startswith("style_encoder") or n.startswith("phoneme_embedding"),
        "cross kv": lambda n: "_ref" in n or "_text" in n,
    }
    for gname, sel in groups.items():
        for name in ("bf16x1", "fp16x1", "fp16 A2.W1", "fp16 A1.W2"):
            fa, fw = MODES[name]
            print(f"  only {gname:14s} {name:12s} {rel_l2(run(fa, fw, sel), x0):.3e}")


def study_codec(seed):
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), seed))
    lat = torch.randn(1, 20, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = CO.decode(wd, lat, DEFAULT_CODEC)
    orig = CO._block

    def mk(fa, fw):
        cache = {}

        def block(w, p, x, spec):
            c = x.shape[1]
            n = CO._rms_c(x, w[f"{p}.norm.weight"], spec.eps)
            y = CO._causal_conv(n, w[f"{p}.mixer.weight"][:, None, :], w[f"{p}.mixer.bias"], groups=c)
            x = x + w[f"{p}.gamma"][None, :, None] * y
            n = CO._rms_c(x, w[f"{p}.ffn_norm.weight"], spec.eps).transpose(1, 2)
            for k in ("w1", "w2"):
                if (p, k) not in cache:
                    cache[(p, k)] = fw(w[f"{p}.ffn.{k}.weight"])
            h = torch.nn.functional.gelu(fa(n) @ cache[(p, "w1")].t() + w[f"{p}.ffn.w1.bias"])
            y = (fa(h) @ cache[(p, "w2")].t() + w[f"{p}.ffn.w2.bias"]).transpose(1, 2)
            return x + w[f"{p}.ffn_gamma"][None, :, None] * y
        return block

    print("codec decode SNR vs exact fp32 oracle (1 x 20 frames; FFN products only, resampling convs exact)")
    for name, (fa, fw) in MODES.items():
        if fa is None:
            continue
        CO._block = mk(fa, fw)
        try:
            with torch.no_grad():
                got = CO.decode(wd, lat, DEFAULT_CODEC)
        finally:
            CO._block = orig
        print(f"  FFNs {name:12s} {snr_db(got, ref):.1f} dB")


if __name__ == "__main__":
    with open(os.path.join(ROOT, "tests", "golden", "meta.json")) as f:
        seed = json.load(f)["weights_seed"]
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    torch.set_num_threads(8)
    if what in ("dit", "both"):
        study_dit(seed)
    if what in ("codec", "both"):
        study_codec(seed)
