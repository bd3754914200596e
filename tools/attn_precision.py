"""Latent error and time of the default preset with the attention operands at fp16 vs split-bf16 (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from oracle import dit_oracle as O
from smalltts_amd.engine import HipEngine
from smalltts_amd.weights import dit_param_specs, synth_state_dict
from tests.conftest import rel_l2

SEED = 1234
sd = O.to_torch(synth_state_dict(dit_param_specs(), SEED))
dev = torch.device("cuda", 0)
cases = [(8, 75, 15, 30), (2, 75, 38, 128), (1, 225, 64, 198), (1, 1, 1, 1), (3, 40, 7, 11)]
refs = []
for (B, N, R, P) in cases:
    g = torch.Generator().manual_seed(100 + N)
    ref = torch.randn(B, R, 64, generator=g); ids = torch.randint(1, 198, (B, P), generator=g)
    rl = torch.full((B,), R); pm = torch.ones(B, P, dtype=torch.bool); mask = torch.ones(B, N, dtype=torch.bool)
    noise = torch.randn(4, B, N, 64, generator=g)
    with torch.no_grad():
        ox = O.sample_dmd(sd, O.encode_conditions(sd, ref, rl, ids, pm), pm, mask, noise, 4)
    refs.append((ref, rl, ids, pm, mask, noise, ox.numpy()))
for prec in ("f16", "f16,attn=bf16x3", "f16,attn=bf16", "bf16x3", "bf16x3,attn=f16"):
    eng = HipEngine(0, prec); eng.load_synthetic(SEED, parts=("dit", "decoder")); eng.finalize()
    errs = []
    for (ref, rl, ids, pm, mask, noise, ox) in refs:
        x = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, noise=noise).cpu().numpy()
        errs.append(rel_l2(x, ox))
    inp = bench.make_inputs(dev, 0)
    for _ in range(3): bench.one_step(eng, inp, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): bench.one_step(eng, inp, 2 + i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{prec:20s} latent rel-L2 " + "  ".join(f"{c}: {e:.2e}" for c, e in zip(cases, errs)) + f"   | one batch at a time {ms:.2f} ms", flush=True)
    eng.close()
