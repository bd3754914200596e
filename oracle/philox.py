"""ORACLE (test infrastructure): numpy restatement of the sampler's noise generator.
Philox4x32-10 (Salmon et al., SC'11) counter-based RNG + Box-Muller, element i drawn from
counter (i // 4 lo, i // 4 hi, stream lo, stream hi), key (seed lo, seed hi).  The reference
draws its noise from numpy's unseeded global RNG (infer/onnx.py:104), so this generator is
build-defined; parity tests of the sampler inject noise explicitly."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, seed: int) -> np.ndarray:
    """ctr: (n, 4) uint32 counters -> (n, 4) uint32."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, -1).astype(np.uint32)


def philox_randn(n: int, seed: int, stream: int = 0) -> np.ndarray:
    q = (n + 3) // 4
    idx = np.arange(q, dtype=np.uint64)
    ctr = np.stack([idx & MASK, idx >> np.uint64(32), np.full(q, stream & 0xFFFFFFFF, np.uint64),
                    np.full(q, (stream >> 32) & 0xFFFFFFFF, np.uint64)], -1).astype(np.uint32)
    r = philox4x32_10(ctr, seed)
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    out = np.empty((q, 4), np.float32)
    for p in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * p]))
        th = np.float32(6.283185307179586) * u[:, 2 * p + 1]
        out[:, 2 * p] = rad * np.cos(th)
        out[:, 2 * p + 1] = rad * np.sin(th)
    return out.reshape(-1)[:n]
