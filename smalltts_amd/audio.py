"""Host-side audio helpers for the CLI surface: WAV read/write (PCM_16 out, like the reference's
`sf.write(..., 24000, subtype="PCM_16")`, tryme.py:29) and the Kaiser-windowed sinc resampler the
clone path uses (reference infer/utils.py:7-23: torchaudio Resample, sinc_interp_kaiser,
lowpass_filter_width=1024, rolloff=0.94, beta=14.769656459379492).  torchaudio/soundfile are not
installed here, so both are restated from their published algorithms (resampler parity unpinned)."""
from __future__ import annotations

import math
import struct
from typing import Tuple

import numpy as np

RESAMPLE_WIDTH = 1024
RESAMPLE_ROLLOFF = 0.94
RESAMPLE_BETA = 14.769656459379492


def write_wav_pcm16(path: str, audio: np.ndarray, sample_rate: int = 24_000) -> None:
    """float [-1,1] -> 16-bit PCM; clamp then scale by 32767 (reference server audio.rs:22-37)."""
    a = np.asarray(audio, dtype=np.float32).reshape(-1)
    pcm = np.round(np.clip(a, -1.0, 1.0) * 32767.0).astype("<i2")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (float32 samples (frames,) or (frames, channels), sample_rate). PCM 8/16/24/32 and float32/64."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data, "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data[: len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = np.frombuffer(data, "<i4").astype(np.float32) / float(1 << 31)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(data, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    if ch > 1:
        x = x[: len(x) // ch * ch].reshape(-1, ch)
    return x, sr


def _sinc_kernel(orig: int, new: int):
    base = min(orig, new) * RESAMPLE_ROLLOFF
    width = int(math.ceil(RESAMPLE_WIDTH * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -RESAMPLE_WIDTH, RESAMPLE_WIDTH)
    window = np.i0(RESAMPLE_BETA * np.sqrt(1.0 - (t / RESAMPLE_WIDTH) ** 2)) / np.i0(RESAMPLE_BETA)
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0.0, 1.0, np.sin(t) / t)
    return (k * window * (base / orig)).astype(np.float32), width


def resample_hq(x: np.ndarray, sr: int, target: int) -> np.ndarray:
    """(samples,) or (channels, samples) float -> resampled to `target` Hz (polyphase windowed sinc)."""
    if sr == target:
        return np.asarray(x, dtype=np.float32)
    x = np.asarray(x, dtype=np.float32)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[None]
    g = math.gcd(int(sr), int(target))
    orig, new = sr // g, target // g
    kern, width = _sinc_kernel(orig, new)                      # (new, 2*width + orig)
    length = x.shape[-1]
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    klen = kern.shape[1]
    n_frames = (xp.shape[-1] - klen) // orig + 1
    out = np.empty((x.shape[0], n_frames, new), dtype=np.float32)
    # strided frames (stride = orig) x polyphase bank; processed in chunks to bound memory
    win = np.lib.stride_tricks.sliding_window_view(xp, klen, axis=-1)[:, ::orig][:, :n_frames]
    step = max(1, (1 << 24) // klen)
    for s in range(0, n_frames, step):
        out[:, s:s + step] = win[:, s:s + step] @ kern.T
    y = out.reshape(x.shape[0], -1)[:, : int(math.ceil(new * length / orig))]
    return y[0] if squeeze else y
