#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec of the full synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch per GPU: condition-encode -> 4-step DMD sampler
-> codec decode of B=8 utterances x 10 s (N=75 frames, R=15 reference frames, P=30 phoneme ids:
BASELINE.json configs[1], workload of the reference's src/server/src/bin/bench.rs:3-25), inputs
already resident in HBM, synthetic seeded weights (no released weights exist offline).  With N GPUs
every rank synthesises its own 8-utterance shard (weak scaling, no data-path collective) and the
waveform shards are reassembled with one RCCL all-gather inside the timed step.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     : dominant kernel, algorithmic work / HIP-event time measured live on the launch stream
  cpu_baseline : the CPU oracle (stand-in for the reference's ORT-CPU path, which cannot run offline)
                 timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, N_FRAMES, R_FRAMES, P_TOK, DMD_STEPS = 8, 75, 15, 30, 4
AUDIO_SEC_PER_UTT = N_FRAMES * 3200 / 24000.0  # 10.0
SEED = 20260928
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16


def make_inputs(device, rank, r_frames=R_FRAMES, p_tok=P_TOK):
    g = torch.Generator().manual_seed(1000 + rank)
    ref = torch.randn(B, r_frames, 64, generator=g)
    ids = (torch.arange(p_tok) % 197 + 1)[None].repeat(B, 1)                     # bench.rs:6: tokens 1..P (wrapped into the 198-symbol table)
    d = dict(ref=ref, ref_len=torch.full((B,), r_frames, dtype=torch.int64), ids=ids,
             ph_mask=torch.ones(B, p_tok, dtype=torch.bool), mask=torch.ones(B, N_FRAMES, dtype=torch.bool))
    t = torch.arange(48000, dtype=torch.float32) / 24000.0                       # bench.rs:8-13: 2 s, 440 Hz unit sine
    d["ref_wav"] = torch.sin(2 * np.pi * 440.0 * t)[None, None].repeat(B, 1, 1)
    d["ref3"] = torch.cat([ref, ref, torch.zeros_like(ref)], 0)                   # CFG rows (distill.py:76-99)
    d["len3"] = torch.cat([d["ref_len"], d["ref_len"], torch.zeros_like(d["ref_len"])], 0)
    d["ids3"] = torch.cat([ids, torch.zeros_like(ids), ids], 0)
    d["pm3"] = torch.cat([d["ph_mask"], torch.zeros_like(d["ph_mask"]), d["ph_mask"]], 0)
    return {k: v.to(device) for k, v in d.items()}


WORKLOADS = {
    "dmd4": "cond-encode + 4-step DMD sampler + codec decode",                                  # BASELINE configs[1]
    "clone": "codec encode of a 2 s reference wav + cond-encode + 4-step DMD + codec decode",   # configs[2]
    "teacher128": "cond-encode (3B CFG rows) + 128-step teacher ODE with CFG + codec decode",   # configs[4]
}


def one_step(eng, inp, seed, ctx=None, gather=None, workload="dmd4", pcm16=False):
    """One pass of the hot path over this rank's batch of 8; with several ranks the waveform shards are reassembled by
    smalltts_amd.parallel.ShardContext.gather_waveforms (ONE all-gather: RCCL over xGMI) inside the step."""
    if workload == "clone":      # reference bench.rs times the codec encode of the 2 s / 440 Hz sine in every call
        ref = eng.codec_encode(inp["ref_wav"])
        cache = eng.cond_encode(ref, inp["ref_len"], inp["ids"], inp["ph_mask"])
        x = eng.sample(cache, inp["mask"], num_steps=DMD_STEPS, seed=seed)
    elif workload == "teacher128":
        cache = eng.cond_encode(inp["ref3"], inp["len3"], inp["ids3"], inp["pm3"])
        x = eng.sample(cache, inp["mask"], num_steps=128, mode="ode", cfg=True, seed=seed)
    else:
        cache = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
        x = eng.sample(cache, inp["mask"], num_steps=DMD_STEPS, seed=seed)
    audio = eng.codec_decode(x)
    if ctx is not None and ctx.collective:
        if pcm16:                # the CLIs write PCM_16 (tryme.py:29): gathering int16 halves the bytes on the links
            audio = eng.pcm16(audio)
        return ctx.gather_waveforms(audio, ctx.world * B, out=gather)
    return audio


def kernel_source_hash():
    """Hash of the kernel sources: profiles/*_latest files record the hash they were measured on, so a stale counter file
    (kernels changed since the rocprofv3 passes) is detected instead of silently quoted."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smalltts_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_probe(candidates, budget_s=20.0):
    """One denoiser call of the bench shape at each candidate thread count (bounded: counts above 16 are skipped once a count
    takes more than `budget_s` or is 1.5x slower than the best so far) -> {threads: seconds}."""
    from oracle import dit_oracle as O
    from smalltts_amd.weights import dit_param_specs, synth_state_dict
    w = O.to_torch(synth_state_dict(dit_param_specs(), SEED))
    g = torch.Generator().manual_seed(1000)
    ref = torch.randn(B, R_FRAMES, 64, generator=g)
    ids = torch.arange(1, P_TOK + 1)[None].repeat(B, 1)
    pm = torch.ones(B, P_TOK, dtype=torch.bool)
    mask = torch.ones(B, N_FRAMES, dtype=torch.bool)
    x = torch.randn(B, N_FRAMES, 64, generator=g)
    out = {}
    with torch.no_grad():
        torch.set_num_threads(min(candidates))
        cache = O.encode_conditions(w, ref, torch.full((B,), R_FRAMES), ids, pm)
        for c in candidates:
            torch.set_num_threads(c)
            best = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                O.denoise_step(w, x, mask, torch.full((B,), 0.5), cache, ph_mask=pm)
                best = min(best, time.perf_counter() - t0)
                if best > budget_s:
                    break
            out[c] = best
            if best > budget_s or best > 1.5 * min(out.values()):
                break
    return out


def cpu_baseline(cores):
    """Oracle on the host cores over the WHOLE 8 x 10 s batch: cond-encode + 4 DMD steps + codec decode of all eight utterances
    in one batched call (round 3 decoded one utterance and scaled by eight: VERDICT r3).  Best of two passes after one warm-up of
    the DiT part, ~20-30 s of CPU work.  kind = "port": the reference's ORT path cannot run offline."""
    from oracle import codec_oracle as CO
    from oracle import dit_oracle as O
    from smalltts_amd.weights import DEFAULT_CODEC, codec_decoder_param_specs, dit_param_specs, synth_state_dict
    torch.set_num_threads(cores)
    w = O.to_torch(synth_state_dict(dit_param_specs(), SEED))
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), SEED))
    g = torch.Generator().manual_seed(1000)
    ref = torch.randn(B, R_FRAMES, 64, generator=g)
    ids = torch.arange(1, P_TOK + 1)[None].repeat(B, 1)
    pm = torch.ones(B, P_TOK, dtype=torch.bool)
    mask = torch.ones(B, N_FRAMES, dtype=torch.bool)
    noise = torch.randn(DMD_STEPS, B, N_FRAMES, 64, generator=g)
    t_dit = t_dec = float("inf")
    with torch.no_grad():
        cache = O.encode_conditions(w, ref, torch.full((B,), R_FRAMES), ids, pm)   # warm-up: page faults, thread pool
        x = O.sample_dmd(w, cache, pm, mask, noise, DMD_STEPS)
        for _ in range(2):
            t0 = time.perf_counter()
            cache = O.encode_conditions(w, ref, torch.full((B,), R_FRAMES), ids, pm)
            x = O.sample_dmd(w, cache, pm, mask, noise, DMD_STEPS)
            t_dit = min(t_dit, time.perf_counter() - t0)
            t0 = time.perf_counter()
            CO.decode(wd, x, DEFAULT_CODEC)
            t_dec = min(t_dec, time.perf_counter() - t0)
            if t_dec > 25.0:   # a slow host: one pass is the bounded sample
                break
    total = t_dit + t_dec
    return {"value": round(B * AUDIO_SEC_PER_UTT / total, 3), "unit": "audio-seconds/sec", "cores": cores,
            "kind": "port",
            "sample": f"CPU oracle (torch fp32) on the whole 8x10s batch: cond-encode + 4 DMD steps ({t_dit:.2f} s) + codec decode of all "
                      f"8 utterances in one call ({t_dec:.2f} s), best of 2 passes; stands in for the reference's ORT-CPU path, which "
                      "cannot run offline",
            "dit_seconds": round(t_dit, 3), "codec_decode_seconds": round(t_dec, 3)}


# SURVEY 8(d) / BASELINE.md 4 algorithmic work per 8 x 10 s batch (R = 15, P = 30): flops counted once, bf16 weights read once
# per use, activations negligible.  Codec: this build's CodecSpec: 13.5 GMAC per audio-second; bytes = weights
# once (2 B / parameter) + one read and one write of every stage-boundary image.
def codec_decode_algo_bytes(spec=None, batch=B, frames=N_FRAMES):
    """SURVEY 8(d) bytes of one codec decode: one write + one read of every stage-boundary tensor (the fp32 image a stage hands
    to the next: stem output, each ConvTranspose output, the waveform) + every decoder weight once at 2 B / parameter.
    Derived from the CodecSpec in code; DEFAULT_CODEC at 8 x 75 frames: 2 x 0.94 GB + 0.69 GB = 2.57 GB."""
    from smalltts_amd.weights import DEFAULT_CODEC, codec_decoder_param_specs
    spec = spec or DEFAULT_CODEC
    n_stage = len(spec.dec_depths)
    t, boundary = frames, 0
    for i in range(n_stage):
        if i > 0:
            t *= spec.ratios[i - 1]
        boundary += batch * t * (spec.n_filters << (n_stage - 1 - i)) * 4
    boundary += batch * t * 4                                   # the waveform
    weights = sum(int(np.prod(shape)) for _, shape, *_ in codec_decoder_param_specs(spec)) * 2
    return 2 * boundary + weights


ALGO = {
    "dit_sampler": {"flops": 695e9, "bytes": 1.722e9},
    "cond_encoders": {"flops": 38e9, "bytes": 0.225e9},
    "codec_decode": {"flops": 2.16e12, "bytes": None},   # bytes: codec_decode_algo_bytes() (filled in main: needs the package)
}


def ensure_world(n_gpus, argv):
    """`--gpus N` must describe the job that runs.  Under a launcher (RANK / WORLD_SIZE set) a mismatch is an error; without one,
    N > 1 re-executes this script under the driver's own command line (python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...) — it never prints a world-1 line for N > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if n_gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if launched:
        if world != n_gpus:
            sys.exit(f"bench.py: --gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks; run\n  python -m "
                     f"torch.distributed.run --nnodes=1 --nproc-per-node {n_gpus} --master-addr 127.0.0.1 --master-port 29500 "
                     f"bench.py --gpus {n_gpus} ...")
        return
    if n_gpus == 1:
        return
    shared = os.environ.get("SMTTS_DIST_BACKEND") == "gloo"     # tests: several gloo ranks may share one GPU
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n_gpus and not (shared and ndev >= 1):
        sys.exit(f"bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs, found {ndev} (one process per GPU over RCCL); "
                 f"refusing to print a line for fewer GPUs than asked")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stderr.write("bench.py: --gpus %d without a launcher: re-executing as\n  %s\n" % (n_gpus, " ".join(cmd)))
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def _stamped(path_json, ksha):
    """Committed rocprofv3 summaries are quoted only when they were measured on THESE kernel sources (hash stamp)."""
    if not os.path.exists(path_json):
        return None
    with open(path_json) as f:
        j = json.load(f)
    return j if j.get("kernel_src_sha") == ksha else {"stale": j.get("kernel_src_sha")}


def profile_block(eng, inp, args, tuning, shapes=True):
    """Per-kernel HIP-event timing (events on the launch stream) of `reps` untimed passes of the workload under `tuning`,
    one batch at a time -> {"roofline", "phase_roofline", "kernel_breakdown"} for that tuning."""
    prev = eng.set_tuning(tuning)
    try:
        return _profile_block(eng, inp, args, tuning, shapes)
    finally:
        eng.set_tuning(prev)


def _profile_block(eng, inp, args, tuning, shapes):
    reps = min(args.steps, 5)
    eng.profile(True, tagged=True)   # names come back as "<phase>/<kernel>" (enc, mod, dit, dec.s<i>, cenc.s<i>)
    for i in range(reps):
        one_step(eng, inp, 900 + i, None, None, args.workload)
    torch.cuda.synchronize()
    tagged = eng.profile_report()
    eng.profile(False)
    merged, phases, dec_launches = {}, {}, {}
    for r in tagged:
        ph, _, kname = r["name"].partition("/") if "/" in r["name"] else ("-", "", r["name"])
        k = merged.setdefault(kname, {"name": kname, "ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "bytes8d": 0.0})
        group = "dit_sampler" if ph in ("dit", "mod") else "cond_encoders" if ph == "enc" else \
                "codec_encode" if ph.startswith("cenc") else "codec_decode"   # untagged: head conv / stem of the decoder
        g_ = phases.setdefault(group, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "bytes8d": 0.0})
        for d_ in (k, g_):
            d_["ms"] += r["ms"]; d_["flops"] += r["flops"]; d_["bytes"] += r["bytes"]; d_["bytes8d"] += r.get("bytes8d", r["bytes"])
        k["launches"] += r["launches"]
        if group == "codec_decode":
            dec_launches[kname] = dec_launches.get(kname, 0) + r["launches"]
    rows = list(merged.values())
    ksha = kernel_source_hash()
    suffix = "" if tuning == "throughput" else "_latency"     # profiles/kernel_stats_latest.csv = the timed (throughput) tuning
    traffic = _stamped(os.path.join(ROOT, "profiles", f"traffic{suffix}_latest.json"), ksha)
    by_traffic = traffic.get("by_prof_name", {}) if traffic and "stale" not in traffic else {}
    # DiT and codec roofline fractions separately.  "8d" = the survey's own accounting (ALGO: flops once, bf16
    # weights once; codec: stage-boundary images once each way + weights once) over the sum of the phase's kernel times;
    # "as_run" = the operand bytes the kernels are given in this precision (fp16 = 2 B, split-bf16 = 4 B per element, split-K
    # partials included).
    pr = {}
    for g, v in sorted(phases.items(), key=lambda kv: -kv[1]["ms"]):
        ms = v["ms"] / reps
        a = ALGO.get(g, {})
        fl8 = a.get("flops") if args.workload == "dmd4" and a.get("flops") else v["flops"] / reps
        by8 = a.get("bytes") if args.workload == "dmd4" and a.get("bytes") else v["bytes8d"] / reps
        pr[g] = {"ms_per_step": round(ms, 3),
                 "8d": {"TFLOPs": round(fl8 / ms / 1e9, 2), "GBs": round(by8 / ms / 1e6, 1), "bytes": round(by8),
                        "mfma_frac": round(fl8 / ms / 1e9 / MFMA_BF16_PEAK_TF, 5), "hbm_frac": round(by8 / ms / 1e6 / HBM_PEAK_GBS, 5)},
                 "as_run": {"TFLOPs": round(v["flops"] / reps / ms / 1e9, 2), "GBs": round(v["bytes"] / reps / ms / 1e6, 1),
                            "hbm_frac": round(v["bytes"] / reps / ms / 1e6 / HBM_PEAK_GBS, 5)}}
    if "codec_decode" in pr and by_traffic and args.workload == "dmd4":
        # counter traffic of the decoder's kernels per batch (per-class average x this phase's launches) / the 8(d) bytes
        seen = {k: n for k, n in dec_launches.items() if k in by_traffic}
        tr = sum(by_traffic[k] * n for k, n in seen.items()) / reps
        pr["codec_decode"]["counter_traffic_bytes"] = round(tr)
        pr["codec_decode"]["wasted_traffic"] = round(tr / ALGO["codec_decode"]["bytes"], 3)
        pr["codec_decode"]["traffic_kernels_covered"] = f"{len(seen)} of {len(dec_launches)}"
    rows.sort(key=lambda r: -r["ms"])
    tot = sum(r["ms"] for r in rows)
    top = rows[0]
    per_ev = top["ms"] / top["launches"] * 1e-3
    # rocprofv3 average of the same kernel class from the committed, hash-stamped summary measured under THIS tuning
    # (tools/profile_round.sh); the HIP-event pair reads 1.5-3 us high on 10-20 us kernels
    rocprof_us, rocprof_src = None, None
    spath = os.path.join(ROOT, "profiles", f"kernel_stats{suffix}_latest.csv")
    meta = _stamped(os.path.join(ROOT, "profiles", f"kernel_stats{suffix}_latest.meta.json"), ksha)
    if meta and "stale" not in meta and os.path.exists(spath):
        import csv
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from prof_names import prof_name
        us = calls = 0.0
        with open(spath) as f:
            for r in csv.DictReader(f):
                if prof_name(r["kernel"]) == top["name"]:
                    us += float(r["total_us"])
                    calls += float(r["calls"])
        if calls:
            rocprof_us = us / calls
            rocprof_src = f"profiles/kernel_stats{suffix}_latest.csv ({meta.get('tag', '')}: {meta.get('cmd', '')})"
    elif meta and "stale" in meta:
        rocprof_src = f"stale: profiles/kernel_stats{suffix}_latest.csv was measured on kernel sources {meta['stale']}, not quoted"

    def fracs(per):
        tf = top["flops"] / top["launches"] / per / 1e12
        gbs8 = top["bytes8d"] / top["launches"] / per / 1e9
        return tf, gbs8, tf / MFMA_BF16_PEAK_TF, gbs8 / HBM_PEAK_GBS
    tf_e, gbs_e, mf_e, hf_e = fracs(per_ev)
    per = rocprof_us * 1e-6 if rocprof_us else per_ev
    tf, gbs8, mfma_frac, hbm_frac = fracs(per)
    gbs_run = top["bytes"] / top["launches"] / per / 1e9
    bound = "mfma" if mfma_frac >= hbm_frac else "hbm"
    roof = {
        "kernel": top["name"], "tuning": tuning, "bound": bound,
        "achieved": round(tf if bound == "mfma" else gbs8, 3),
        "peak": MFMA_BF16_PEAK_TF if bound == "mfma" else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round(max(mfma_frac, hbm_frac), 5), "traffic": None,
        "frac_source": "rocprofv3 average launch duration (" + rocprof_src + ")" if rocprof_us else
                       "HIP events on the launch stream (no hash-matching rocprofv3 summary for this tuning under profiles/)",
        "frac_events": round(max(mf_e, hf_e), 5), "avg_launch_us_events": round(per_ev * 1e6, 3),
        "rocprof_avg_us": round(rocprof_us, 3) if rocprof_us else None,
        "avg_launch_us": round(per * 1e6, 3), "launches_per_step": top["launches"] // reps,
        "share_of_kernel_time": round(top["ms"] / tot, 4),
        "flops_per_launch": round(top["flops"] / top["launches"]), "bytes8d_per_launch": round(top["bytes8d"] / top["launches"]),
        "as_run_bytes_per_launch": round(top["bytes"] / top["launches"]), "as_run_GBs": round(gbs_run, 1),
        "mfma_frac": round(mfma_frac, 5), "hbm_frac": round(hbm_frac, 5), "kernel_src_sha": ksha,
        "note": "SURVEY 8(d) accounting: algorithmic flops (2MNK, counted once) and algorithmic bytes per launch / the kernel's "
                "average launch duration; as_run_* = the operand + output bytes the launch is given",
    }
    if rocprof_src and not rocprof_us:
        roof["rocprof_source"] = rocprof_src
    # HBM bytes per launch of that kernel from the PMC passes of tools/profile_round.sh (committed under profiles/)
    if traffic and "stale" not in traffic:
        roof["traffic"] = by_traffic.get(top["name"])
        roof["traffic_source"] = f"profiles/traffic{suffix}_latest.json (" + str(traffic.get("tag", "")) + ")"
        if roof["traffic"] and top["bytes8d"]:
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / (top["bytes8d"] / top["launches"]), 3)
    elif traffic:
        roof["traffic_source"] = f"stale: profiles/traffic{suffix}_latest.json was measured on kernel sources {traffic['stale']}, not quoted"
    if shapes:
        # the dominant class broken down by product shape (profile mode 3: class names carry M x N x K): what its average hides
        eng.profile(True, shapes=True)
        for i in range(reps):
            one_step(eng, inp, 950 + i, None, None, args.workload)
        torch.cuda.synchronize()
        shaped = eng.profile_report()
        eng.profile(False)
        by_shape = {}
        for r in shaped:
            kname = r["name"].partition("/")[2] if "/" in r["name"] else r["name"]
            cls_, _, shp = kname.partition(" ")
            if cls_ == top["name"] and shp:
                a_ = by_shape.setdefault(shp, {"ms": 0.0, "launches": 0, "flops": 0.0})
                a_["ms"] += r["ms"]; a_["launches"] += r["launches"]; a_["flops"] += r["flops"]
        table = []
        for shp, a_ in sorted(by_shape.items(), key=lambda kv: -kv[1]["ms"])[:3]:
            us = a_["ms"] / a_["launches"] * 1e3
            tfs = a_["flops"] / a_["launches"] / us / 1e6
            table.append({"MxNxK": shp, "launches_per_step": a_["launches"] // reps, "avg_us_events": round(us, 2),
                          "TFLOPs": round(tfs, 1), "mfma_frac": round(tfs / MFMA_BF16_PEAK_TF, 4)})
        if table:
            roof["by_shape"] = table
    breakdown = [
        {"name": r["name"], "launches_per_step": r["launches"] // reps, "ms_per_step": round(r["ms"] / reps, 4),
         "TFLOPs": round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2), "GBs_8d": round(r["bytes8d"] / max(r["ms"], 1e-9) / 1e6, 1),
         "GBs_as_run": round(r["bytes"] / max(r["ms"], 1e-9) / 1e6, 1)}
        for r in rows[:16]]
    return {"roofline": roof, "phase_roofline": pr, "kernel_breakdown": breakdown}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default sized so the timed region is >= 3 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="f16",
                    help="f16 (default: one fp16 MFMA per product on the block / encoder / codec-FFN GEMMs, split-bf16 on the "
                         "conditioning and in / out projections), bf16x3 (split-bf16 everywhere), bf16; site overrides as f16,cond=f16")
    ap.add_argument("--workload", default="dmd4", choices=list(WORKLOADS),
                    help="dmd4 = the headline configuration; clone / teacher128 = BASELINE.json configs[2] / configs[4]")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent batches in flight on one GPU: batch i runs, whole, on HIP stream i %% IN_FLIGHT with its own "
                         "workspace (default 3; 1 = one batch at a time on one stream)")
    ap.add_argument("--no-pipeline", dest="in_flight", action="store_const", const=1, help="same as --in-flight 1")
    ap.add_argument("--gather", default="f32", choices=["f32", "pcm16"], help="dtype of the N > 1 waveform all-gather")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the timed K-step region until this much has been timed")
    ap.add_argument("--max-repeats", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tuning", default="auto", choices=["auto", "latency", "throughput"],
                    help="engine tuning of the timed region: auto = throughput with batches in flight, latency with one at a time; "
                         "forcing it lets tools/profile_round.sh trace the throughput-tuned kernels one batch at a time")
    ap.add_argument("--no-sequential", action="store_true", help="skip the one-batch-at-a-time leg (value_sequential)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (the other single-GPU configurations timed behind the headline in the same run: "
                         "clone = BASELINE configs[2], teacher128 = configs[4], dmd4 at R = 38 / P = 128 = SURVEY 8(d)'s secondary point)")
    args = ap.parse_args()
    ensure_world(args.gpus, sys.argv[1:])

    from smalltts_amd.parallel import ShardContext
    ctx = ShardContext.from_env()          # one process per GPU (torch.distributed.run); single process when WORLD_SIZE <= 1
    world, rank, device = ctx.world, ctx.rank, ctx.device
    n_gpus = world
    assert n_gpus == args.gpus or os.environ.get("SMTTS_DIST_FORCE") == "1", (n_gpus, args.gpus)   # ensure_world() settled this
    torch.cuda.set_device(ctx.device_index)

    from smalltts_amd.engine import HipEngine
    eng = HipEngine(ctx.device_index, args.precision)
    secondary = not args.no_secondary and args.workload == "dmd4" and n_gpus == 1
    eng.load_synthetic(SEED, parts=("dit", "decoder", "encoder") if args.workload == "clone" or secondary else ("dit", "decoder"))
    eng.finalize()
    # (tests: a world-1 run can stand in for rank r of a larger job — same inputs as that rank's shard)
    inp = make_inputs(device, int(os.environ.get("SMTTS_BENCH_RANK_SEED", rank)))
    pcm16 = args.gather == "pcm16"
    gdtype = torch.int16 if pcm16 else torch.float32
    samples = 3200 * N_FRAMES
    barrier = ctx.barrier

    def run_steps(n, seed0, in_flight, tuning=None, workload=None, inp_=None):
        """n full passes of the hot path, each over its own batch of 8.  With in_flight > 1 consecutive batches are issued
        round-robin to that many HIP streams (one workspace each): the latency-bound phases of one batch (condition
        encoders, DiT: grids of 30-190 workgroups on 256 CUs) fill the CUs another batch's kernels leave idle.  Every batch
        still goes through the whole path inside the timed region; nothing is shared between batches but the weights."""
        workload = workload or args.workload
        inp_ = inp if inp_ is None else inp_
        if in_flight <= 1:
            out = None
            prev = eng.set_tuning(tuning) if tuning else None
            try:
                for i in range(n):
                    out = one_step(eng, inp_, seed0 + i, ctx, gathers[0] if gathers else None, workload, pcm16)
            finally:
                if tuning:
                    eng.set_tuning(prev)
            return out
        cur = torch.cuda.current_stream(device)
        for s_ in streams[:in_flight]:
            s_.wait_stream(cur)
        prev = eng.set_tuning(tuning or "throughput")   # unsplit GEMMs, no engine side stream, capped persistent grids: fewest CU-us per kernel
        out = None
        try:
            for i in range(n):
                with torch.cuda.stream(streams[i % in_flight]):
                    eng.use_workspace(f"batch{i % in_flight}")
                    out = one_step(eng, inp_, seed0 + i, ctx, gathers[i % in_flight] if gathers else None, workload, pcm16)
        finally:
            eng.use_workspace(None)
            eng.set_tuning(prev)
        for s_ in streams[:in_flight]:
            cur.wait_stream(s_)
        return out

    in_flight = max(1, args.in_flight)
    timed_tuning = args.tuning if args.tuning != "auto" else ("throughput" if in_flight > 1 else "latency")
    streams = [torch.cuda.Stream(device) for _ in range(in_flight)] if in_flight > 1 else []
    gathers = [ctx.gather_buffer(world * B, samples, gdtype) for _ in range(in_flight)] if ctx.collective else []  # one per slot
    run_steps(args.warmup, 0, in_flight, timed_tuning)
    # The timed region is EXACTLY K steps between barrier + device sync on both sides, MAX over ranks.  The driver fixes K (20
    # steps = 0.2 s here), so the region is REPEATED until >= --min-seconds have been timed in all (every rank takes the same
    # decision: the elapsed time it looks at is the max over ranks); the line reports the MEDIAN repeat, min / max beside it.
    dts = []
    out = None
    while True:
        barrier()
        t0 = time.perf_counter()
        out = run_steps(args.steps, 100 + 1000 * len(dts), in_flight, timed_tuning)
        barrier()
        dts.append(ctx.max_over_ranks(time.perf_counter() - t0))
        if sum(dts) >= args.min_seconds or len(dts) >= args.max_repeats:
            break
    dt = sorted(dts)[len(dts) // 2]
    # the same K steps one batch at a time on one stream: the latency of a batch, and the strict reading of "at batch = 8"
    ns = max(3, min(args.steps, 100))
    dt_seq = None
    if not args.no_sequential:
        run_steps(2, 50, 1, "latency")
        barrier()
        t1 = time.perf_counter()
        run_steps(ns, 60, 1, "latency")
        barrier()
        dt_seq = ctx.max_over_ranks(time.perf_counter() - t1)
    assert torch.isfinite(out.float()).all()

    # ---- the other single-GPU configurations, timed in the same run behind the headline ----------------
    # Same protocol per leg: warm-up, then K steps between barrier + device sync, in flight (throughput tuning) and one batch at a
    # time (latency tuning).  K is small (the whole block takes a few seconds): these are driver-clocked points, not the headline.
    sec = None
    if secondary:
        sec = {}
        legs = [("clone", "clone", inp, max(3, min(args.steps, 12))),
                ("dmd4_R38_P128", "dmd4", make_inputs(device, rank, 38, 128), max(3, min(args.steps, 12))),
                ("teacher128", "teacher128", inp, 3)]
        for name, wl, inp2, k in legs:
            torch.cuda.synchronize()
            eng.release_workspaces()          # shapes differ per leg: fresh per-slot scratch, nothing in flight while it is swapped
            run_steps(min(in_flight, k) if in_flight > 1 else 1, 7000, in_flight, timed_tuning, wl, inp2)
            barrier()
            t2 = time.perf_counter()
            o2 = run_steps(k, 7100, in_flight, timed_tuning, wl, inp2)
            barrier()
            d_if = ctx.max_over_ranks(time.perf_counter() - t2)
            ks = 2 if wl == "teacher128" else k
            run_steps(1, 7200, 1, "latency", wl, inp2)
            barrier()
            t2 = time.perf_counter()
            o2 = run_steps(ks, 7300, 1, "latency", wl, inp2)
            barrier()
            d_sq = ctx.max_over_ranks(time.perf_counter() - t2)
            assert torch.isfinite(o2.float()).all()
            a_s = B * AUDIO_SEC_PER_UTT
            r_, p_ = (38, 128) if name == "dmd4_R38_P128" else (R_FRAMES, P_TOK)
            sec[name] = {"workload": WORKLOADS[wl] + f", B=8 x 10 s (N=75, R={r_}, P={p_})", "steps": k, "steps_sequential": ks,
                         "batches_in_flight": in_flight, "ms_per_step": round(1e3 * d_if / k, 3), "value": round(a_s * k / d_if, 2),
                         "rtf": round(d_if / (a_s * k), 7), "sequential_ms_per_step": round(1e3 * d_sq / ks, 3),
                         "value_sequential": round(a_s * ks / d_sq, 2), "unit": "audio-seconds/sec"}
        torch.cuda.synchronize()
        eng.release_workspaces()

    audio_s = n_gpus * B * AUDIO_SEC_PER_UTT * args.steps
    audio_s_seq = n_gpus * B * AUDIO_SEC_PER_UTT * ns
    res = {
        "metric": "audio-seconds/sec (RTF) at batch=8, 10 s utterances; 1->8 GPU scaling",
        "value": round(audio_s / dt, 2), "unit": "audio-seconds/sec", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16": "fp16 MFMA single pass on block/encoder/codec-FFN GEMMs + split-bf16 (x3) MFMA on conditioning and in/out "
                         "projections; fp32 accumulate, residual stream, norms, softmax, sampler state",
                  "bf16x3": "bf16 MFMA x3 split (fp32-class), fp32 accumulate/residual",
                  "bf16": "bf16 MFMA single pass, fp32 accumulate/residual"}.get(args.precision, args.precision),
        "data": "synthetic (seeded inputs + seeded random weights; no released weights offline)",
        "rtf": round(dt / audio_s, 7),
        # one batch of 8 at a time on one stream (no batches in flight): the strict reading of the metric's "at batch = 8"
        "value_sequential": round(audio_s_seq / dt_seq, 2) if dt_seq else None,
        "sequential_ms_per_step": round(1e3 * dt_seq / ns, 3) if dt_seq else None,
        "rtf_sequential": round(dt_seq / audio_s_seq, 7) if dt_seq else None,
        # the K-step region repeated: value / ms_per_step are the median repeat
        "timed_seconds": round(sum(dts), 3), "repeats": len(dts), "ms_per_step_min": round(1e3 * min(dts) / args.steps, 3),
        "ms_per_step_max": round(1e3 * max(dts) / args.steps, 3), "spread": round((max(dts) - min(dts)) / dt, 4),
        "dist": {"backend": ctx.backend, "collective": bool(ctx.collective), "world": world},
        "config": {"workload": WORKLOADS[args.workload] + ", B=8 x 10 s per GPU "
                               "(N=75 frames, R=15 ref frames, P=30 tokens; reference bench.rs workload)",
                   "global_batch": n_gpus * B, "utterance_seconds": AUDIO_SEC_PER_UTT, "sampler_steps": 128 if args.workload == "teacher128" else DMD_STEPS,
                   "parallelism": (f"dp{n_gpus}: one process per GPU, 8-utterance shard each, one {args.gather} waveform all-gather "
                                   f"({ctx.backend})") if n_gpus > 1 else "single GPU",
                   "batches_in_flight": in_flight, "tuning": timed_tuning, "precision": args.precision,
                   # sites the fp16 range guard / calibration moved to split-bf16 on these weights (none on the seeded recipe):
                   # a published number states the precision actually in force
                   "precision_demoted_sites": eng.precision_in_force()["demoted"]},
    }

    if sec is not None:
        res["secondary"] = sec
    if os.environ.get("SMTTS_BENCH_DUMP_ROWS") == "1":
        # tests: one checksum per utterance of the last timed step's (gathered) batch — int64 sum of the fp32 bit patterns — so
        # that a world-N line can be compared row block by row block with world-1 runs (tests/test_bench_gpu.py)
        o_ = out.reshape(out.shape[0], -1)
        bits = o_.view(torch.int16 if o_.dtype == torch.int16 else torch.int32).to(torch.int64)
        res["row_checksums"] = [int(v) for v in bits.sum(dim=1).cpu()]
    if rank == 0 and not args.no_roofline:
        ALGO["codec_decode"]["bytes"] = float(codec_decode_algo_bytes())
        # The roofline block describes the configuration that was TIMED: its per-kernel passes run under the timed region's
        # tuning (throughput when batches are in flight), one batch at a time on one stream (include/smalltts_hip.h: the
        # profiler covers one call at a time).  A second block under latency tuning sits next to value_sequential.
        res["roofline_tuning"] = timed_tuning
        blk = profile_block(eng, inp, args, timed_tuning)
        res["phase_roofline"], res["roofline"], res["kernel_breakdown"] = blk["phase_roofline"], blk["roofline"], blk["kernel_breakdown"]
        if timed_tuning != "latency" and not args.no_sequential:
            blk2 = profile_block(eng, inp, args, "latency", shapes=False)
            res["roofline_sequential"] = blk2["roofline"]
            res["phase_roofline_sequential"] = {k: {"ms_per_step": v["ms_per_step"], "mfma_frac_8d": v["8d"]["mfma_frac"],
                                                    "hbm_frac_8d": v["8d"]["hbm_frac"]} for k, v in blk2["phase_roofline"].items()}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and args.workload == "dmd4":
        try:
            avail = len(os.sched_getaffinity(0))
        except Exception:
            avail = os.cpu_count() or 1
        # BASELINE.md 3 asks for all host cores with the count stated.  Graphs of small torch ops stop scaling long before a big
        # host's core count and then collapse (measured on the 256-core GPU box, profiles/r02c_bench.json: 0.05 audio-s/s with
        # 256 threads against 7.0 with 16 — 25 minutes for the leg), so the thread count is chosen by a bounded probe: one
        # denoiser call at every candidate count, the full leg (10-30 s) at the fastest; the probe timings are reported.
        cands = sorted({c for c in (8, 16, 32, 64, avail) if c <= avail})
        probe = cpu_probe(cands)
        best = min(probe, key=probe.get)
        full = cpu_baseline(best)
        full["cores_available"] = avail
        full["thread_probe_ms"] = {str(k): round(v * 1e3, 1) for k, v in probe.items()}
        res["cpu_baseline"] = full
    if rank == 0:
        print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
