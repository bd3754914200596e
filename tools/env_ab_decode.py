"""Full-size codec decode under two values of one environment switch, in one process (the switches are read when an engine is
created): are the outputs bit-identical?   python tools/env_ab_decode.py NAME A B [batch] [frames] [tuning]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
name, va, vb = sys.argv[1:4]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
N = int(sys.argv[5]) if len(sys.argv) > 5 else 75
tuning = sys.argv[6] if len(sys.argv) > 6 else "latency"
lat = torch.randn(B, N, 64, generator=torch.Generator().manual_seed(21)).cuda()
outs = []
for v in (va, vb):
    os.environ[name] = v
    eng = HipEngine(0)
    eng.load_synthetic(5, parts=("decoder",)); eng.finalize()
    eng.set_tuning(tuning)
    o = eng.codec_decode(lat).cpu().numpy()
    rep = np.array_equal(o, eng.codec_decode(lat).cpu().numpy())
    print(f"{name}={v}: finite {np.isfinite(o).all()}  repeatable {rep}")
    outs.append(o)
    eng.close()
a, b = outs
d = (a != b)
snr = 10 * np.log10((b.astype(np.float64) ** 2).sum() / max(((a.astype(np.float64) - b) ** 2).sum(), 1e-300))
print(f"equal: {np.array_equal(a, b)}  differing samples {int(d.sum())} of {d.size}  max |diff| {float(np.abs(a - b).max()):.3e}  snr {snr:.1f} dB")
