"""debug: engine filled by load_synthetic vs engine filled from the numpy recipe (set_tensor): tensors and codec decode"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine
from smalltts_amd.weights import CodecSpec, codec_decoder_param_specs, synth_state_dict
from oracle import codec_oracle as CO
from oracle.dit_oracle import to_torch
SPEC = CodecSpec(n_filters=8, ratios=(8, 5, 5, 4, 2, 2), dec_depths=(1, 1, 1, 1, 1, 1, 1))
specs = codec_decoder_param_specs(SPEC)
sd = synth_state_dict(specs, 11)
a = HipEngine(0, "bf16x3"); a.load_synthetic(11, parts=("decoder",), codec_spec=SPEC); a.finalize()
b = HipEngine(0, "bf16x3"); b.set_codec_spec(SPEC); b.load_state_dict(sd); b.finalize()
bad = 0
for n, sh in specs:
    ta = a.get_tensor(n, sh); tb = b.get_tensor(n, sh)
    if not np.array_equal(ta, tb) or not np.array_equal(tb, sd[n]):
        bad += 1
        if bad < 8: print("DIFF", n, sh, float(np.abs(ta - sd[n]).max()), float(np.abs(tb - sd[n]).max()))
print("tensors differing:", bad, "of", len(specs))
lat = torch.randn(1, 9, 64, generator=torch.Generator().manual_seed(0))
xa = a.codec_decode(lat).cpu().numpy(); xb = b.codec_decode(lat).cpu().numpy()
xa2 = a.codec_decode(lat).cpu().numpy()
with torch.no_grad():
    ref = CO.decode(to_torch(sd), lat, SPEC).numpy()
snr = lambda g, r: 10 * np.log10((r.astype(np.float64) ** 2).sum() / max(((g.astype(np.float64) - r) ** 2).sum(), 1e-300))
print("a vs b equal:", np.array_equal(xa, xb), "a repeat equal:", np.array_equal(xa, xa2), "snr a/b", snr(xa, xb), "a vs oracle", snr(xa, ref), "b vs oracle", snr(xb, ref))
