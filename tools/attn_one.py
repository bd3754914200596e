#!/usr/bin/env python
"""Run the DiT attention kernel at the bench shape repeatedly (for rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, "bf16x3")
B, N, H, dh, R, P = 8, 75, 8, 120, 15, 30
g = torch.Generator().manual_seed(0)
qkvg = torch.randn(B, N, 4 * H * dh, generator=g)
w = torch.ones(H, dh)
inv = 1.0 / (1e4 ** (torch.arange(0, 64, 2).float() / 64))
rope = (torch.arange(N).float()[:, None] * inv[None]).repeat_interleave(2, -1).contiguous()
kr, vr = torch.randn(B, H, R, dh, generator=g), torch.randn(B, H, R, dh, generator=g)
kt, vt = torch.randn(B, H, P, dh, generator=g), torch.randn(B, H, P, dh, generator=g)
for _ in range(20):
    o = eng.test_attention(qkvg, w, w, 1e-6, rope, 64, H, dh, kr, vr, kt, vt)
torch.cuda.synchronize()
print("ok", float(o.abs().mean()))
