"""Stage chain vs one launch per block on the GPU box: where do the two decodes differ?  python tools/chain_check.py [B] [frames]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smalltts_amd.engine import HipEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 75
tuning = sys.argv[3] if len(sys.argv) > 3 else "latency"
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3       # blocks of the C = 32 stage
from smalltts_amd.weights import CodecSpec
spec = CodecSpec(dec_depths=(1, 1, 1, 1, 1, 1, depth)) if depth != 3 or len(sys.argv) > 4 else None
os.environ["SMTTS_CHAIN_MIN"] = "1"
zero = sys.argv[5].split(",") if len(sys.argv) > 5 else []   # e.g. "1.gamma,1.ffn_gamma": tensors of the C = 32 stage's blocks set to zero
lat = torch.randn(B, N, 64, generator=torch.Generator().manual_seed(21)).cuda()
outs = {}
for tag, v in (("chain", "1"), ("blocks", "0")):
    os.environ["SMTTS_STAGE_CHAIN"] = v
    eng = HipEngine(0)
    eng.load_synthetic(5, parts=("decoder",), codec_spec=spec)
    from smalltts_amd.weights import codec_decoder_param_specs, DEFAULT_CODEC
    names = {n: sh for n, sh in codec_decoder_param_specs(spec or DEFAULT_CODEC)}
    for z in zero:
        blk, leaf = z.split(".", 1)
        cand = [n for n in names if n.endswith(f"stages.6.{blk}.{leaf}")]
        assert len(cand) == 1, (z, cand, [n for n in names if ".6." in n][:40])
        eng.set_tensor(cand[0], np.zeros(names[cand[0]], np.float32))
    eng.finalize()
    eng.set_tuning(tuning)
    outs[tag] = eng.codec_decode(lat).cpu().numpy()
    again = eng.codec_decode(lat).cpu().numpy()
    print(tag, "repeatable:", np.array_equal(outs[tag], again), "finite:", np.isfinite(outs[tag]).all())
    eng.close()
a, b = outs["chain"], outs["blocks"]
d = a != b
print("equal:", np.array_equal(a, b), " differing samples:", int(d.sum()), "of", d.size, " max |diff|", float(np.abs(a - b).max()),
      " snr", 10 * np.log10((b.astype(np.float64) ** 2).sum() / max(((a.astype(np.float64) - b) ** 2).sum(), 1e-300)))
for u in range(B):
    idx = np.nonzero(d[u, 0])[0]
    if idx.size == 0:
        print(f"utt {u}: identical"); continue
    tiles = np.unique(idx // 32)
    print(f"utt {u}: {idx.size} samples differ in {tiles.size} tiles of {a.shape[-1] // 32}; first samples {idx[:12].tolist()}  first tiles {tiles[:16].tolist()}  "
          f"sample % 32 histogram {np.bincount(idx % 32, minlength=32).tolist()}")
    big = np.abs(a[u, 0] - b[u, 0])
    j = int(big.argmax())
    print(f"   largest diff {big[j]:.3e} at sample {j} (tile {j // 32}, frame-in-tile {j % 32}); chain {a[u,0,j]:.6f} blocks {b[u,0,j]:.6f}")
