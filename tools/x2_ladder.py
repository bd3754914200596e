"""Codec decode at the precisions of SITE_CODEC_CONV, against the engine's own split-bf16 decode (97.6 dB vs the fp32 oracle,
tests/test_precision_gpu.py — exact for this purpose), plus the decode time of the bench's 8 x 75 frames:
    [SMTTS_X2_MINK=.. SMTTS_X2_MAXK=..] python tools/x2_ladder.py
The K range selects which ConvTranspose stages run the two-pass fp16 product under codec_conv=f16x2 (K = 4096, 2048, 1024, 512 for
decoder stages 1-4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smalltts_amd.engine import HipEngine


def snr_db(got, ref):
    return float(10 * torch.log10((ref.double() ** 2).sum() / ((got.double() - ref.double()) ** 2).sum()))


eng = HipEngine(0, "bf16x3")
eng.load_synthetic(1, parts=("decoder",)); eng.finalize()
lat = torch.randn(8, 75, 64, generator=torch.Generator().manual_seed(11)).cuda()
ref = eng.codec_decode(lat).clone()
tag = f"X2 K range [{os.environ.get('SMTTS_X2_MINK', '512')}, {os.environ.get('SMTTS_X2_MAXK', '1024')}]"
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for p in ("f16", "f16,codec_conv=f16x2", "f16,codec_conv=f16", "bf16x3,codec_conv=f16x2"):
    eng.set_precision(p)
    got = eng.codec_decode(lat).clone()
    again = eng.codec_decode(lat)
    rep = torch.equal(got, again)
    for _ in range(3):
        eng.codec_decode(lat)
    t0.record()
    for _ in range(10):
        eng.codec_decode(lat)
    t1.record(); torch.cuda.synchronize()
    s = [snr_db(got[b], ref[b]) for b in range(8)]
    print(f"{tag}  {p:26s} SNR vs split-bf16 decode: min {min(s):.1f} dB, mean {sum(s) / 8:.1f} dB   decode {t0.elapsed_time(t1) / 10:.3f} ms   repeatable {rep}")
