"""Which GEMM site loses the 1e-3 contract when the weights have outlier rows?  

    python tools/outlier_ladder.py

Seeded DiT weights with 0.5 % of the rows of ff.w1 / ff.w3 and / or to_q / to_k_self scaled (tests/test_range_guard_gpu.py),
4-step DMD latents against the fp32 oracle for: the default preset, the default with ONE site at split-bf16, split-bf16."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import dit_oracle as O
from smalltts_amd.engine import HipEngine, SITES
from tests.test_range_guard_gpu import _inputs, _outlier_dit_weights
from tests.conftest import rel_l2

torch.set_num_threads(16)
ref, rl, ids, pm, mask, noise = _inputs()
for label, ff, qk, lat20 in (("ff x30 only", 30.0, 1.0, False), ("q/k x30 only", 1.0, 30.0, False), ("latent channels x20 only", 1.0, 1.0, True),
                              ("ff x30 + q/k x30 + latents x20", 30.0, 30.0, True), ("ff x100 + q/k x100 + latents x20", 100.0, 100.0, True)):
    sd = _outlier_dit_weights(ff, qk)
    r = ref.clone()
    if not lat20:
        r[:, :, [3, 17, 40]] /= 20.0
    w = O.to_torch(sd)
    with torch.no_grad():
        ox = O.sample_dmd(w, O.encode_conditions(w, r, rl, ids, pm), pm, mask, noise, 4).numpy()
    eng = HipEngine(0)
    eng.load_state_dict(sd)
    eng.finalize()
    print(f"== {label}")
    rows = ["f16"] + [f"f16,{s}=bf16x3" for s in ("dit_block", "attn", "encoder", "cross_kv")] + ["f16,dit_block=bf16x3,attn=bf16x3", "bf16x3"]
    for p in rows:
        eng.set_precision(p)
        x = eng.sample(eng.cond_encode(r, rl, ids, pm), mask, num_steps=4, noise=noise).cpu().numpy()
        sat = {k: v for k, v in eng.saturations().items() if v}
        print(f"  {p:36s} latent rel-L2 {rel_l2(x, ox):.2e}   clamps {sat}")
    eng.close()
