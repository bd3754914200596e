"""Run the headline codec decode (B=8, 75 latent frames) a few times (for rocprofv3 --pmc):  python tools/codec_one.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine
eng = HipEngine(0, sys.argv[2] if len(sys.argv) > 2 else "f16")
eng.load_synthetic(1, parts=("decoder",)); eng.finalize()
lat = torch.randn(8, 75, 64, generator=torch.Generator().manual_seed(4)).cuda()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = eng.codec_decode(lat)
torch.cuda.synchronize()
print("ok", tuple(out.shape))
