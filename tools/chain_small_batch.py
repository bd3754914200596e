"""Stage chain (C = 32: three blocks per tile in one launch, one warm-up tile per run) against one launch per block on SMALL decodes,
where a wave's run is 1-3 tiles and the warm-up tile is 33-100 % extra work.  GPU box:  python tools/chain_small_batch.py"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from smalltts_amd.engine import HipEngine
    eng = HipEngine(0); eng.load_synthetic(bench.SEED, parts=("decoder",)); eng.finalize()
    for B, T in ((1, 15), (1, 38), (1, 75), (2, 75), (4, 75), (8, 75)):
        x = torch.randn(B, T, 64, device="cuda")
        for _ in range(5): eng.codec_decode(x)
        torch.cuda.synchronize(); best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): eng.codec_decode(x)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        print(f"  B={B} T={T:3d} ({B * T * 3200 // 32:6d} tiles at the last stage): {best:.3f} ms per decode")
else:
    for chain in ("1", "0"):
        print(f"SMTTS_STAGE_CHAIN={chain}"); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, SMTTS_STAGE_CHAIN=chain))
