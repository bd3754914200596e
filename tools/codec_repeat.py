import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine
from smalltts_amd.weights import CodecSpec
eng = HipEngine(0, "bf16x3")
eng.load_synthetic(1, parts=("decoder",)); eng.finalize()
lat = torch.randn(1, 75, 64, generator=torch.Generator().manual_seed(4))
a = eng.codec_decode(lat).cpu(); b = eng.codec_decode(lat).cpu()
print("same input twice: equal", torch.equal(a, b), "maxdiff", float((a - b).abs().max()))
lat2 = torch.cat([lat, lat], 0)
c = eng.codec_decode(lat2).cpu()
print("B=2 row0 vs B=1:", torch.equal(c[0], a[0]), float((c[0] - a[0]).abs().max()), " row1 vs row0:", torch.equal(c[1], c[0]), float((c[1]-c[0]).abs().max()))
d = (c[0] - a[0]).abs().squeeze()
nz = torch.nonzero(d > 0).squeeze()
print("first/last differing sample:", int(nz[0]) if nz.numel() else None, int(nz[-1]) if nz.numel() else None, "count", nz.numel())
for fused in (0, 1):
    eng.lib.smtts_test_set_fused_ffn(eng.h, fused)
    a1 = eng.codec_decode(lat).cpu(); c1 = eng.codec_decode(lat2).cpu()
    print("fused", fused, "B1==B2row0:", torch.equal(c1[0], a1[0]), float((c1[0]-a1[0]).abs().max()))
