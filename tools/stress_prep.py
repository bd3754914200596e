"""r03 root-cause session for the fused q / k prep attention kernel (GPU box).
  python tools/stress_prep.py dump  [iters]   cond_encode repeats with the style encoder's attention inputs / outputs dumped per layer:
                                              on a differing repeat, say whether the kernel's INPUT or only its OUTPUT moved, and where
  python tools/stress_prep.py count [iters]   just count differing repeats (variant builds: SMTTS_LIB=...)
  python tools/stress_prep.py neigh [iters]   single-stream cond_encode next to an unrelated neighbour on another stream
Run with SMTTS_ATTN_PREP=1 (or 0 for the control)."""
import os, sys
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "count"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda", 0)
B, R, P, D, L = 8, 15, 30, 512, 12
M = B * R
LAYER = M * 4 * D * 4 + M * D * 2
dump = None
if mode == "dump":
    dump = torch.zeros(L * LAYER, dtype=torch.uint8, device=dev)
    os.environ["SMTTS_DBG_ATTN_PTR"] = hex(dump.data_ptr())

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from smalltts_amd.engine import HipEngine

eng = HipEngine(0); eng.load_synthetic(bench.SEED, parts=("dit", "decoder")); eng.finalize()
inp = bench.make_inputs(dev, 0)
tag = f"PREP={os.environ.get('SMTTS_ATTN_PREP', '0')} lib={os.path.basename(os.environ.get('SMTTS_LIB', 'default'))} " \
      f"hwq={os.environ.get('GPU_MAX_HW_QUEUES', '-')} lds={os.environ.get('SMTTS_DBG_ATTN_LDS', '-')}"


def cond():
    c = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"], debug=True)
    out = {k: v.clone() for k, v in c.items() if torch.is_tensor(v)}
    if dump is not None:
        out["_dump"] = dump.clone()
    return out


def layer_views(d, l):
    base = l * LAYER
    qkvg = d[base: base + M * 4 * D * 4].view(torch.float32).view(B, R, 4, 8, 64)     # (b, n, {q,k,v,g}, h, d)
    o = d[base + M * 4 * D * 4: base + LAYER].view(torch.float16).view(B, R, 8, 64)
    return qkvg, o


if mode in ("dump", "count"):
    a = cond(); bad = 0; shown = 0
    for i in range(iters):
        b = cond()
        diff = [k for k in a if not k.startswith("_") and not torch.equal(a[k], b[k])]
        bad += bool(diff)
        if diff and dump is not None and shown < 4:
            shown += 1
            print(f"  repeat {i}: outputs that differ: {diff}")
            for l in range(L):
                qa, oa = layer_views(a["_dump"], l); qb, ob = layer_views(b["_dump"], l)
                qeq, oeq = torch.equal(qa, qb), torch.equal(oa, ob)
                if qeq and oeq:
                    continue
                if not qeq:
                    w = (qa != qb).nonzero()
                    print(f"    layer {l}: attention INPUT differs ({len(w)} elements; parts {sorted(set(w[:, 2].tolist()))}) -> upstream of the kernel")
                else:
                    w = (oa != ob).nonzero()
                    bh = sorted(set((int(x[0]), int(x[2])) for x in w))
                    ns = sorted(set(int(x[1]) for x in w))
                    ds = sorted(set(int(x[3]) for x in w))
                    err = (oa.float() - ob.float()).abs().max().item()
                    print(f"    layer {l}: input identical, OUTPUT differs: {len(w)} elements, (b, h) = {bh[:12]}{'...' if len(bh) > 12 else ''}, "
                          f"queries {ns}, dims {ds[:8]}..{ds[-1]} ({len(ds)}), max |diff| {err:.3e}, "
                          f"nan {torch.isnan(ob.float()).sum().item()}")
                break
    print(f"[{tag}] cond_encode dual-stream: {bad} of {iters} repeats differ")

if mode == "neigh":
    eng.set_dual_stream(False)
    side = torch.cuda.Stream(dev)
    big = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    vec = torch.randn(64 << 20, device=dev)
    lat = torch.randn(8, 75, 64, device=dev)

    def neighbour(kind):
        with torch.cuda.stream(side):
            if kind == "mm":
                for _ in range(12):
                    torch.mm(big, big)
            elif kind == "ew":
                for _ in range(12):
                    vec.mul_(1.0001)
            elif kind == "codec":
                eng.use_workspace("side")
                eng.codec_decode(lat)
                eng.use_workspace(None)
            elif kind == "cond":
                eng.use_workspace("side")
                eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
                eng.use_workspace(None)

    a = cond()
    for kind in ("none", "ew", "mm", "codec", "cond"):
        bad = 0
        for i in range(iters):
            side.wait_stream(torch.cuda.current_stream(dev))
            neighbour(kind)
            b = cond()
            torch.cuda.current_stream(dev).wait_stream(side)
            bad += any(not torch.equal(a[k], b[k]) for k in a)
        torch.cuda.synchronize()
        print(f"[{tag}] single-stream cond_encode next to '{kind}' on another stream: {bad} of {iters} differ")

if mode == "aggr":
    # single-stream cond_encode next to ONE kind of synthetic neighbour (tools/ubench/aggressors.hip) on another stream
    import ctypes as C
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "bin", "libaggr.so"))
    lib.aggr_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int]
    eng.set_dual_stream(False)
    side = torch.cuda.Stream(dev)
    src = torch.randn(64 << 20, device=dev)
    sink = torch.zeros(1024, device=dev)
    names = ["LDS-DMA", "ds_read + mfma", "v_exp_f32", "ds_write + barrier", "global_load", "v_pk_fma_f32", "DPP", "mfma",
             "v_permlane32_swap", "gemm3-like k-loop"]
    its = [60000, 120000, 600000, 60000, 60000, 600000, 600000, 120000, 600000, 60000]
    a = cond()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for kind in range(10):
        with torch.cuda.stream(side):   # how long one aggressor launch runs alone (it must outlast a cond_encode, ~1.5 ms)
            t0.record(); lib.aggr_launch(kind, C.c_void_p(side.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(sink.data_ptr()), its[kind], src.numel(), 1024, 32768); t1.record()
        torch.cuda.synchronize()
        print(f"   '{names[kind]}' aggressor alone: {t0.elapsed_time(t1):.2f} ms per launch")
        for lds in (32768,):
            bad = 0
            for i in range(iters):
                side.wait_stream(torch.cuda.current_stream(dev))
                rc = lib.aggr_launch(kind, C.c_void_p(side.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(sink.data_ptr()),
                                     its[kind], src.numel(), 1024, lds)
                assert rc == 0, rc
                b = cond()
                torch.cuda.current_stream(dev).wait_stream(side)
                bad += any(not torch.equal(a[k], b[k]) for k in a)
            torch.cuda.synchronize()
            print(f"[{tag}] single-stream cond_encode next to the '{names[kind]}' aggressor: {bad} of {iters} differ", flush=True)
