"""Uninitialised-scratch hunt: the sampler / encoders must give the same bits whatever the scratch buffer held before.
    python tools/debug_precision_default.py      (GPU box)
Fills the engine's scratch with 0x00 / 0xFF / 0x7F bytes between runs, per precision preset and per phase."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
torch.set_num_threads(16)
from smalltts_amd.engine import HipEngine

seed = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "meta.json")))["weights_seed"]


def inputs(B=8, N=75, R=15, P=30, s=0):
    gen = torch.Generator().manual_seed(s)
    ref = torch.randn(B, R, 64, generator=gen); ids = torch.arange(1, P + 1)[None].repeat(B, 1)
    noise = torch.randn(4, B, N, 64, generator=gen)
    return ref, torch.full((B,), R), ids, torch.ones(B, P, dtype=torch.bool), torch.ones(B, N, dtype=torch.bool), noise


ref, rl, ids, pm, mask, noise = inputs()
eng = HipEngine(0); eng.load_synthetic(seed, parts=("dit", "decoder")); eng.finalize()


def fill(byte):
    torch.cuda.synchronize()
    if eng._ws is not None:
        eng._ws.fill_(byte)
    torch.cuda.synchronize()


def snapshot(debug):
    c = eng.cond_encode(ref, rl, ids, pm, debug=debug)
    return {k: t.clone() for k, t in c.items() if torch.is_tensor(t)}


eng.set_precision("f16")
a = snapshot(True)
for i in range(24):
    b = snapshot(True)
    for k in ("ref_seq", "phoneme_mem", "k_ref", "k_text"):
        d = (a[k].float() - b[k].float()).abs()
        if float(d.max()) > 0:
            nz = (d > 0)
            idx = nz.nonzero()
            dims = [sorted(set(idx[:, j].tolist())) for j in range(idx.shape[1])]
            print(f"repeat {i} {k} shape {tuple(d.shape)}: {int(nz.sum())} elements differ, max {float(d.max()):.2e}; index sets per dim (first 12): {[x[:12] for x in dims]} sizes {[len(x) for x in dims]}")
print("done")
