"""ORACLE (test infrastructure, NOT product code): Kaiser-windowed sinc resampler of the clone path.

Reference call site: `src/smalltts/infer/utils.py:7-23` — `torchaudio.transforms.Resample(sr, 24000,
resampling_method="sinc_interp_kaiser", lowpass_filter_width=1024, rolloff=0.94, beta=14.769656459379492)`.
The arithmetic lives in torchaudio 2.8.0 (`uv.lock:1869-1870`), which is not installed here and not
vendored by the reference: PARITY UNPINNED.  This file restates torchaudio's published algorithm
(`torchaudio.functional._get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`) in its closed form:

    o, n   = orig / gcd, new / gcd;  base = min(o, n) * rolloff;  width = ceil(lpw * o / base)
    y[m]   = (base / o) * sum_j x[j] * sinc_pi(t) * kaiser(t),   f = m // n, p = m % n,
             t = clamp(((j - f*o) / o - p / n) * base, -lpw, lpw),   j - f*o in [-width, width + o)
    kaiser(t) = I0(beta * sqrt(1 - (t / lpw)^2)) / I0(beta);  len(y) = ceil(n * len(x) / o)

evaluated directly per (output sample, input sample) pair in float64 — no polyphase bank, no framing — so it
shares no code and no data layout with the product's bank builder (`smalltts_amd/audio.py:_sinc_kernel`) or
the device kernel (`smtts_resample_poly`).  Only tests import it.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import i0

LPW, ROLLOFF, BETA = 1024, 0.94, 14.769656459379492  # infer/utils.py:9-16


def resample(x: np.ndarray, sr: int, target: int, chunk: int = 2048) -> np.ndarray:
    """(samples,) float -> (ceil(new * samples / orig),) float64."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if sr == target:
        return x.copy()
    g = math.gcd(int(sr), int(target))
    o, n = sr // g, target // g
    base = min(o, n) * ROLLOFF
    width = int(math.ceil(LPW * o / base))
    n_out = int(math.ceil(n * x.size / o))
    y = np.zeros(n_out)
    rel = np.arange(-width, width + o, dtype=np.float64)          # j - f*o
    i0b = i0(BETA)
    for s in range(0, n_out, chunk):
        m = np.arange(s, min(s + chunk, n_out))
        f, p = m // n, m % n
        t = (rel[None, :] / o - (p / n)[:, None]) * base
        t = np.clip(t, -LPW, LPW)
        win = i0(BETA * np.sqrt(np.maximum(0.0, 1.0 - (t / LPW) ** 2))) / i0b
        tp = t * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            k = np.where(tp == 0.0, 1.0, np.sin(tp) / tp) * win * (base / o)
        j = f[:, None] * o + rel[None, :].astype(np.int64)
        ok = (j >= 0) & (j < x.size)
        y[m] = (np.where(ok, x[np.clip(j, 0, x.size - 1)], 0.0) * k).sum(-1)
    return y
