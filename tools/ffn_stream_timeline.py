"""Per-phase timeline of one pass of codec_ffn_stream (GPU box, debug build -DFS_TIMELINE loaded through SMTTS_LIB):

    make -C smalltts_amd/csrc BUILD=build_tl LIB=../libsmalltts_hip_tl.so EXTRA=-DFS_TIMELINE
    SMTTS_LIB=smalltts_amd/libsmalltts_hip_tl.so python tools/ffn_stream_timeline.py

Wave 0 of every workgroup stamps the shader clock (s_memtime, calibrated against s_memrealtime) through its first three passes:
   pass start | tile loaded + normalised | per ring step: own DMA pieces landed / barrier passed / next slot issued / MFMAs + GELU issued |
   ring steps done | residual added + stores issued | stores drained
The decoder is cut off behind the stage under test (a CodecSpec whose LAST stage is C = 128 resp. 256 at the bench batch's row
count), so the stamps in the buffer are those of that stage's last block."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from smalltts_amd.engine import HipEngine
from smalltts_amd.weights import CodecSpec

PASSES, N = 3, 160
for Cw, spec in ((128, CodecSpec(n_filters=128, ratios=(8, 5, 5, 4), dec_depths=(1, 1, 1, 1, 3))),
                 (256, CodecSpec(n_filters=256, ratios=(8, 5, 5), dec_depths=(1, 1, 1, 3)))):
    eng = HipEngine(0, "f16")
    lib = eng.lib
    lib.smtts_debug_read_fs_timeline.argtypes = [C.c_void_p, C.c_int]
    eng.load_synthetic(1, parts=("decoder",), codec_spec=spec)
    eng.finalize()
    lat = torch.randn(8, 75, 64, generator=torch.Generator().manual_seed(4)).cuda()
    for _ in range(2):
        eng.codec_decode(lat)
    torch.cuda.synchronize()
    eng.profile(True)
    assert lib.smtts_debug_clear_fs_timeline() == 0
    eng.codec_decode(lat)
    torch.cuda.synchronize()
    rows = {r["name"]: r for r in eng.profile_report()}
    eng.profile(False)
    us = rows[f"codec_ffn_stream<{Cw}>"]["ms"] * 1e3 / rows[f"codec_ffn_stream<{Cw}>"]["launches"]
    buf = np.zeros(256 * PASSES * N, np.uint64)
    assert lib.smtts_debug_read_fs_timeline(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, PASSES, N).astype(np.int64)
    NT1 = 4 * Cw // 32
    nstep = NT1 + 2
    M = 8 * 75 * spec.hop
    nw = 8 if Cw == 128 else 4
    print()
    print(f"== codec_ffn_stream<{Cw}>  M = {M} rows ({M // (nw * 32)} passes of {nw * 32} frames over 256 workgroups), {us:.1f} us per launch by events")
    for p in range(PASSES):
        tp = t[:, p]
        live = (tp[:, 0] > 0) & (tp[:, 142] > 0)
        if not live.any():
            continue
        tp = tp[live]
        tick = float(np.median((tp[:, 151] - tp[:, 150]) * 10.0 / np.maximum(tp[:, 142] - tp[:, 0], 1)))   # ns per shader tick
        us_ = lambda a: a * tick / 1e3
        med = lambda a: float(np.median(us_(a)))
        own = np.stack([tp[:, 2 + 4 * i] - (tp[:, 5 + 4 * (i - 1)] if i else tp[:, 1]) for i in range(nstep)], 1)
        bar = np.stack([tp[:, 3 + 4 * i] - tp[:, 2 + 4 * i] for i in range(nstep)], 1)
        iss = np.stack([tp[:, 4 + 4 * i] - tp[:, 3 + 4 * i] for i in range(nstep)], 1)
        cmp_ = np.stack([tp[:, 5 + 4 * i] - tp[:, 4 + 4 * i] for i in range(nstep)], 1)
        total = tp[:, 142] - tp[:, 0]
        load = tp[:, 1] - tp[:, 0]
        ring = tp[:, 140] - tp[:, 1]
        epi = tp[:, 141] - tp[:, 140]
        drain = tp[:, 142] - tp[:, 141]
        start = us_(tp[:, 0] - tp[:, 0].min())
        print(f"  pass {p}: {live.sum()} workgroups, shader clock {1e3 / tick:.0f} MHz; start spread p10-p90 {np.percentile(start, 10):.1f}-{np.percentile(start, 90):.1f} us")
        print(f"    per pass (median): total {med(total):.2f} us = tile load + norm {med(load):.2f} | {nstep} ring steps {med(ring):.2f} | residual add + store issue {med(epi):.2f} | drain {med(drain):.2f}")
        s = lambda a: sum(float(np.median(us_(a[:, i]))) for i in range(nstep))
        print(f"    ring steps, summed medians: wait own DMA {s(own):.2f} | wait barrier {s(bar):.2f} | issue next slot {s(iss):.2f} | fragment reads + MFMAs + GELU {s(cmp_):.2f}")
        show = list(range(nstep)) if nstep <= 18 else list(range(6)) + list(range(nstep - 6, nstep))
        print("      step:        " + " ".join(f"{i:5d}" for i in show))
        for lab, a in (("wait own DMA", own), ("wait barrier", bar), ("issue next", iss), ("MFMA + GELU", cmp_)):
            print(f"      {lab:12s} " + " ".join(f"{float(np.median(us_(a[:, i]))):5.2f}" for i in show))
    eng.close()
