"""Study (test infrastructure): can the codec FFN's GELU run in PACKED fp16?

The one-pass block kernels (C = 32 / 64) are VALU-bound: ~27 VALU-cycle equivalents per hidden value, 9 + one transcendental of
them the GELU (GeluQ5 in fp32, then one rounding to the fp16 operand of the second product).  In packed fp16 (v_pk_fma_f16: two
values per lane-instruction) the same formula costs ~5.5 + one transcendental.  What it costs in audio is measured here, on the CPU,
by emulating the arithmetic inside the codec oracle's FFN at the default precision (operands of both products rounded to fp16):

  fp32   : h -> GeluQ5 in fp32 -> round to fp16                           (what the product does today)
  f16    : h -> round to fp16 -> |.|, Horner (5 fma), exp2, max, fma all rounded to fp16 after every instruction
  f16/32 : h -> round to fp16 -> the three high-order Horner steps in fp16, the last two + exp2 + result in fp32 -> round
  in16   : h -> round to fp16 FIRST, then GeluQ5 in fp32 -> round           (isolates the cost of the early input rounding)

    python -m tests.studies.gelu_f16_packed
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import codec_oracle as CO
from oracle import dit_oracle as O
from smalltts_amd.weights import DEFAULT_CODEC, codec_decoder_param_specs, synth_state_dict

Q = [-1.000138521194458, -1.1501970291137695, -0.46113213896751404, -0.050880067050457, 0.006735440343618393, -0.00042676751036196947]


def h16(x):
    return x.to(torch.float16).to(torch.float32)


def fma16(a, b, c):
    """fp16 fma: the product of two fp16 values is exact in fp32; one rounding to fp16 at the end (double rounding through fp32
    is below an fp16 ulp / 2^13)."""
    return h16(a * b + c)


def gelu_q5_fp32(x):
    a = x.abs()
    q = torch.full_like(a, Q[5])
    for k in (4, 3, 2, 1, 0):
        q = q * a + Q[k]
    return torch.clamp_min(x, 0) - a * torch.exp2(q)


def gelu_q5_f16(x):
    x = h16(x)
    a = x.abs()
    q = torch.full_like(a, float(np.float16(Q[5])))
    for k in (4, 3, 2, 1, 0):
        q = fma16(q, a, float(np.float16(Q[k])))
    e = h16(torch.exp2(q))
    return fma16(-a, e, torch.clamp_min(x, 0))


def gelu_q5_mixed(x):
    x = h16(x)
    a = x.abs()
    q = torch.full_like(a, float(np.float16(Q[5])))
    for k in (4, 3, 2):
        q = fma16(q, a, float(np.float16(Q[k])))
    q = q * a + Q[1]
    q = q * a + Q[0]
    return torch.clamp_min(x, 0) - a * torch.exp2(q)


# round 5: the packed-fp16 evaluation is limited by its own Horner roundings, so a SHORTER exponent polynomial costs next to nothing:
# degree-3 / degree-4 minimax fits of the same form (tests/studies/gelu_q5_fit.py fit(3), fit(4)), evaluated in fp16 like "f16"
Q3 = [-1.008443832397461, -1.113277792930603, -0.5132908821105957, -0.02108863927423954]
Q4 = [-1.0013247728347778, -1.1435197591781616, -0.47347283363342285, -0.04111006110906601, 0.0033284714445471764]


def gelu_pk_f16_deg(coef):
    def g(x):
        x = h16(x)
        a = x.abs()
        q = torch.full_like(a, float(np.float16(coef[-1])))
        for k in range(len(coef) - 2, -1, -1):
            q = fma16(q, a, float(np.float16(coef[k])))
        e = h16(torch.exp2(q))
        return fma16(-a, e, torch.clamp_min(x, 0))
    return g


GELUS = {"fp32": gelu_q5_fp32, "f16": gelu_q5_f16, "f16/32": gelu_q5_mixed, "in16": lambda x: gelu_q5_fp32(h16(x)),
         "f16 d4": gelu_pk_f16_deg(Q4), "f16 d3": gelu_pk_f16_deg(Q3)}


def block(w, p, x, spec, gelu, max_c):
    c = x.shape[1]
    n = CO._rms_c(x, w[f"{p}.norm.weight"], spec.eps)
    y = CO._causal_conv(n, w[f"{p}.mixer.weight"][:, None, :], CO._opt(w, f"{p}.mixer.bias"), groups=c)
    x = x + CO._scale(w, f"{p}.gamma", y)
    n = CO._rms_c(x, w[f"{p}.ffn_norm.weight"], spec.eps).transpose(1, 2)
    b1, b2 = CO._opt(w, f"{p}.ffn.w1.bias"), CO._opt(w, f"{p}.ffn.w2.bias")
    h = h16(n) @ h16(w[f"{p}.ffn.w1.weight"]).t()
    h = h if b1 is None else h + b1
    h = h16((gelu if c <= max_c else gelu_q5_fp32)(h))
    y = h @ h16(w[f"{p}.ffn.w2.weight"]).t()
    y = (y if b2 is None else y + b2).transpose(1, 2)
    return x + CO._scale(w, f"{p}.ffn_gamma", y)


def decode(w, latents, spec, gelu, max_c):
    x = latents.transpose(1, 2)
    x = CO._causal_conv(x, w["codec.decoder.stem.weight"], CO._opt(w, "codec.decoder.stem.bias"))
    for i in range(spec.n_stages):
        if i > 0:
            r = spec.ratios[i - 1]
            t_in = x.shape[-1]
            y = F.conv_transpose1d(x, w[f"codec.decoder.up.{i}.weight"], CO._opt(w, f"codec.decoder.up.{i}.bias"), stride=r)
            x = y[..., : t_in * r]
        for j in range(spec.dec_depths[i]):
            x = block(w, f"codec.decoder.stages.{i}.{j}", x, spec, gelu, max_c)
    return CO._causal_conv(x, w["codec.decoder.head.weight"], CO._opt(w, "codec.decoder.head.bias"))


def snr_db(a, b):
    a, b = a.double(), b.double()
    return float(10 * torch.log10((b ** 2).sum() / ((a - b) ** 2).sum()))


def main():
    torch.set_num_threads(8)
    spec = DEFAULT_CODEC
    x = torch.linspace(-12, 12, 2000001)
    exact = 0.5 * x.double() * (1 + torch.erf(x.double() / np.sqrt(2)))
    print("max |gelu error| on [-12, 12]  (fp16 rounding of the result itself: 2^-11 relative)")
    for name, g in GELUS.items():
        e = (g(x).double() - exact).abs()
        rel = (e / exact.abs().clamp_min(1e-3)).max()
        print(f"  {name:7s} abs {float(e.max()):.2e}   rel (|gelu| > 1e-3) {float(rel):.2e}")
    w = O.to_torch(synth_state_dict(codec_decoder_param_specs(spec), 20260928))
    lat = torch.randn(1, 20, 64, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = CO.decode(w, lat, spec)
        print("decode SNR vs the fp32 oracle, 1 x 20 frames, FFN operands fp16 (ConvTranspose / stem exact):")
        for max_c, label in ((4096, "every stage"), (64, "C <= 64 only (the one-pass block kernels)"), (256, "C <= 256 (+ the streamed kernels)")):
            for name, g in GELUS.items():
                if name == "fp32" and max_c != 4096:
                    continue
                print(f"  GELU {name:7s} on {label:45s} {snr_db(decode(w, lat, spec, g, max_c), ref):.1f} dB")


if __name__ == "__main__":
    main()
