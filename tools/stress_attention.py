"""Are two attention_mfma kernels (fused prep / separate prep) bitwise repeatable while they run next to each other on two streams?  (GPU box)"""
import os, sys, threading, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smalltts_amd.engine import HipEngine

dev = torch.device("cuda", 0)


def case(B, N, H, dh, rot, seed):
    g = torch.Generator().manual_seed(seed)
    D = H * dh
    qkvg = torch.randn(B, N, 4 * D, generator=g)
    qw, kw = 1 + 0.2 * torch.randn(H, dh, generator=g), 1 + 0.2 * torch.randn(H, dh, generator=g)
    inv = 1.0 / (1e4 ** (torch.arange(0, rot, 2).float() / rot))
    rope = (torch.arange(N).float()[:, None] * inv[None]).repeat_interleave(2, -1).contiguous()
    ms = torch.ones(B, N, dtype=torch.bool)
    return dict(qkvg=qkvg, qw=qw, kw=kw, eps=1e-5, rope=rope, rot_dim=rot, H=H, dh=dh, mask_self=ms)


cases = [case(8, 15, 8, 64, 64, 1), case(8, 30, 4, 128, 128, 2)]
engs = [HipEngine(0), HipEngine(0)]
for mode in ("fused", "prep"):
    base = [engs[i].test_attention(mfma=mode, **cases[i]).clone() for i in range(2)]
    bad = [0, 0]
    def worker(i):
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            for _ in range(200):
                out = engs[i].test_attention(mfma=mode, **cases[i])
                bad[i] += not torch.equal(out, base[i])
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"attention {mode}: two streams side by side, 200 calls each: {bad[0]} (dh 64) / {bad[1]} (dh 128) differ from their solo result")
