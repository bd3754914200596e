"""GPU: the product's DEFAULT operand precision ("f16" mixed: one fp16 MFMA per product on the DiT-block / encoder /
cross-KV / codec-FFN GEMMs, split-bf16 on the conditioning chain, the latent in / out projections and the codec
resampling convs) against the fp32 CPU oracle, at the tolerances the north star states:

  latents: rel-L2 < 1e-3 is the contract (BASELINE.json north_star); the tests assert 3e-4 (>= 3x margin);
  audio:   SNR >= 60 dB is the bound this build states (codec parity is unpinned, oracle/codec_oracle.py); the codec tests
           assert 64 dB here.

The tight 1e-4 / 60 dB-vs-96 dB checks of the split-bf16 preset stay in test_dit_gpu.py / test_fullsize_gpu.py: they prove
the kernels' logic; these prove the shipped configuration."""
import numpy as np
import pytest
import torch

from oracle import codec_oracle as CO
from oracle import dit_oracle as O
from smalltts_amd.weights import DEFAULT_CODEC, codec_decoder_param_specs, codec_encoder_param_specs, synth_state_dict
from tests.conftest import golden, rel_l2

pytestmark = pytest.mark.gpu
TOL_F16 = 3e-4     # >= 3x inside the 1e-3 contract
SNR_F16 = 64.0     # dB, decode vs CPU oracle (bound stated by the build: 60 dB)
E2E_SNR = 64.0     # dB, HIP sample -> HIP decode vs oracle sample -> oracle decode (bound stated by the build for the composed path: 60 dB; measured 68.3)


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300))


@pytest.fixture(scope="module")
def eng(golden_seed):
    from smalltts_amd.engine import DEFAULT_PRECISION, HipEngine
    e = HipEngine(0)                       # default precision: what SmallTTS() and bench.py run
    assert e.precision == DEFAULT_PRECISION == "f16"
    e.load_synthetic(golden_seed, parts=("dit", "decoder", "encoder"))
    e.finalize()
    return e


def _bench_inputs(B=8, N=75, R=15, P=30, seed=0):
    gen = torch.Generator().manual_seed(seed)
    ref = torch.randn(B, R, 64, generator=gen)
    ids = torch.arange(1, P + 1)[None].repeat(B, 1)
    noise = torch.randn(4, B, N, 64, generator=gen)
    return ref, torch.full((B,), R), ids, torch.ones(B, P, dtype=torch.bool), torch.ones(B, N, dtype=torch.bool), noise


@pytest.mark.parametrize("case", ["case_small.npz", "case_cfgrows.npz", "case_bench1.npz"])
def test_default_precision_vs_reference_golden(eng, case):
    """velocity of one denoiser call against fixtures made by the reference's own modules (tests/golden/make_golden.py)."""
    g = golden(case)
    cache = eng.cond_encode(g["ref"], g["ref_len"], g["ids"], g["ph_mask"])
    v = eng.denoise_step(g["x_t"], g["mask"], g["t"], cache).cpu().numpy()
    m = g["mask"].astype(bool)
    err = rel_l2(v[m], g["velocity"][m])
    assert err < TOL_F16, f"{case}: velocity rel L2 {err:.3e}"


def test_default_precision_ln_fold_vs_norm_launches_at_the_bench_shape(eng):
    """The LN-fold of the fused sampler (tests/test_dit_gpu.py) at the SHIPPED precision and the bench shape: the operand image is
    fp16(x (1 + scale)) of the un-normalised residual stream, the mean is taken out after the product — both paths must sit at the
    same distance from the fp32 oracle (asserted: 3e-4 each, and the fold no more than 1.25x the norm launches' error)."""
    ref, ref_len, ids, pm, mask, noise = _bench_inputs()
    cache = eng.cond_encode(ref, ref_len, ids, pm)
    x_fold = eng.sample(cache, mask, num_steps=4, noise=noise).cpu().numpy()
    eng.set_ln_fold(False)
    try:
        x_norm = eng.sample(cache, mask, num_steps=4, noise=noise).cpu().numpy()
    finally:
        eng.set_ln_fold(True)
    eng.set_precision("bf16x3")
    try:
        cache3 = eng.cond_encode(ref, ref_len, ids, pm)
        x_ref = eng.sample(cache3, mask, num_steps=4, noise=noise).cpu().numpy()    # fp32-class operands (7e-6 from the oracle)
    finally:
        eng.set_precision("f16")
    e_fold, e_norm = rel_l2(x_fold, x_ref), rel_l2(x_norm, x_ref)
    print(f"\n[ln-fold, f16 mixed, 8 x 75] vs split-bf16: fold {e_fold:.3e}, norm launches {e_norm:.3e}")
    assert e_fold < TOL_F16 and e_norm < TOL_F16 and e_fold < 1.25 * e_norm + 2e-5


def test_default_precision_sampler_vs_reference_golden(eng):
    g = golden("case_sampler4.npz")
    B, N = g["noise"].shape[1:3]
    P = g["ids"].shape[1]
    cache = eng.cond_encode(g["ref"], np.array([g["ref"].shape[1]]), g["ids"], np.ones((B, P), bool))
    x, steps = eng.sample(cache, np.ones((B, N), bool), num_steps=4, noise=g["noise"], return_steps=True)
    for i in range(4):
        err = rel_l2(steps[i].cpu().numpy(), g["x_pred_steps"][i])
        assert err < TOL_F16, f"step {i}: {err:.3e}"


def test_default_precision_at_bench_shape_and_site_ladder(eng, dit_weights):
    """configs[1] of BASELINE.json (B=8 x 10 s, R=15, P=30, 4 DMD steps): the default against the oracle, then the ladder —
    what each GEMM site group costs in latent error when it alone drops from split-bf16 to one fp16 pass, and what happens
    when the sensitive conditioning / in / out sites are dropped too (printed: this is the evidence for the default's site map)."""
    ref, ref_len, ids, pm, mask, noise = _bench_inputs()
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, ref, ref_len, ids, pm)
        ox = O.sample_dmd(dit_weights, oc, pm, mask, noise, 4).numpy()

    def run():
        return eng.sample(eng.cond_encode(ref, ref_len, ids, pm), mask, num_steps=4, noise=noise).cpu().numpy()

    err = rel_l2(run(), ox)
    rows = [("f16 (default)", err)]
    try:
        for site in ("dit_block", "encoder", "cross_kv", "cond", "convpos"):
            eng.set_precision(f"bf16x3,{site}=f16")
            rows.append((f"bf16x3 + {site}=f16", rel_l2(run(), ox)))
        eng.set_precision("f16,convpos=f16")
        rows.append(("f16 + convpos=f16", rel_l2(run(), ox)))
        eng.set_precision("f16,cond=f16,convpos=f16")
        rows.append(("f16 everywhere incl. cond", rel_l2(run(), ox)))
        eng.set_precision("bf16x3")
        rows.append(("bf16x3", rel_l2(run(), ox)))
        eng.set_precision("bf16")
        rows.append(("bf16", rel_l2(run(), ox)))
    finally:
        eng.set_precision("f16")
    print("\n[precision ladder] latent rel-L2 vs fp32 oracle, B=8 N=75 R=15 P=30, 4 steps")
    for k, v in rows:
        print(f"  {k:28s} {v:.3e}")
    assert err < TOL_F16, f"default precision latent rel L2 {err:.3e}"
    d = dict(rows)
    assert d["bf16x3"] < 1e-4 and d["bf16x3 + dit_block=f16"] < TOL_F16


def test_default_precision_range_edges(eng, dit_weights):
    for (B, N, R, P) in [(2, 75, 38, 128), (1, 225, 64, 198), (1, 1, 1, 1)]:
        g = torch.Generator().manual_seed(100 + N)
        ref = torch.randn(B, R, 64, generator=g)
        ids = torch.randint(1, 198, (B, P), generator=g)
        rl = torch.full((B,), R)
        pm = torch.ones(B, P, dtype=torch.bool)
        mask = torch.ones(B, N, dtype=torch.bool)
        if B > 1:
            rl[-1] = max(1, R // 3); pm[-1, P // 2:] = False; ids[-1, P // 2:] = 0; mask[-1, (2 * N) // 3:] = False
        noise = torch.randn(4, B, N, 64, generator=g)
        with torch.no_grad():
            ox = O.sample_dmd(dit_weights, O.encode_conditions(dit_weights, ref, rl, ids, pm), pm, mask, noise, 4)
        x = eng.sample(eng.cond_encode(ref, rl, ids, pm), mask, noise=noise).cpu().numpy()
        m = mask.numpy()
        err = rel_l2(x[m], ox.numpy()[m])
        # A single frame against single-token conditions has nothing to average the fp16 operand roundings over (one query, three
        # keys): 2.8e-4 with split-bf16 attention operands, 3.2e-4 with the fp16 ones the default uses since round 3
        # (profiles/r03e_attention_operand_precision.txt) — held to 4e-4, 2.5x inside the 1e-3 contract; every other shape to 3e-4.
        tol = 4e-4 if N == 1 else TOL_F16
        assert err < tol, f"B={B} N={N} R={R} P={P}: latent rel L2 {err:.3e}"


def test_default_precision_codec_decode_and_ladder(eng, golden_seed):
    """The bench's codec workload (8 x 75 frames): two utterances against the CPU oracle at the default precision, plus the
    ladder of the codec sites on one utterance."""
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), golden_seed))
    lat = torch.randn(8, 75, 64, generator=torch.Generator().manual_seed(11))
    got = eng.codec_decode(lat).cpu()
    with torch.no_grad():
        refs = {b: CO.decode(wd, lat[b:b + 1], DEFAULT_CODEC).numpy() for b in (0, 5)}
    for b, r in refs.items():
        s = snr_db(got[b:b + 1].numpy(), r)
        assert s > SNR_F16, f"utterance {b}: decode SNR {s:.1f} dB at the default precision"
    rows = []
    try:
        for p in ("f16", "f16,codec_conv=f16x2", "f16,codec_conv=f16", "bf16x3,codec_ffn=f16", "bf16x3", "bf16"):
            eng.set_precision(p)
            rows.append((p, snr_db(eng.codec_decode(lat[:1]).cpu().numpy(), refs[0])))
    finally:
        eng.set_precision("f16")
    print("\n[precision ladder] codec decode SNR vs fp32 oracle, 1 x 75 frames")
    for k, v in rows:
        print(f"  {k:28s} {v:.1f} dB")
    lad = dict(rows)
    # the two-pass fp16 ConvTranspose option (A as one fp16 array x fp16 hi + lo weights on the K = 1024 / 512 stages) sits between
    # the default (split-bf16 there) and single-pass fp16 on the K >= 2048 stages, and holds the 66 dB it was built to hold
    assert lad["f16,codec_conv=f16x2"] > 66.0
    assert lad["f16"] >= lad["f16,codec_conv=f16x2"] > lad["f16,codec_conv=f16"]
    # bitwise repeatable and batch-invariant at the default precision too
    again = eng.codec_decode(lat).cpu()
    assert torch.equal(again, got)


def test_default_precision_codec_encode(eng, golden_seed):
    we = O.to_torch(synth_state_dict(codec_encoder_param_specs(DEFAULT_CODEC), golden_seed))
    t = torch.arange(48000) / 24000.0
    audio = (0.5 * torch.sin(2 * np.pi * 440 * t))[None, None].repeat(2, 1, 1)
    audio[1] += 0.05 * torch.randn(48000, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = CO.encode(we, audio, DEFAULT_CODEC).numpy()
    got = eng.codec_encode(audio).cpu().numpy()
    assert got.shape == ref.shape == (2, 15, 64)
    err = rel_l2(got, ref)
    assert err < 1e-3, f"encode latent rel L2 {err:.3e}"


@pytest.mark.parametrize("M,N,K,cfg", [(600, 960, 960, -1), (600, 3840, 960, -1), (257, 130, 192, -1), (75, 64, 2432, 2),
                                       (4800, 512, 2048, 5)])
def test_gemm3_fp16_single_pass_vs_torch(eng, M, N, K, cfg):
    """gemm3 with one fp16 array per operand: against torch on the SAME fp16-rounded operands (fp64 accumulate) the result is
    exact up to fp32 accumulation order; against the unrounded product the error is the fp16 rounding (2^-11 relative)."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    got = eng.test_gemm3(A, W, bias, split=2, cfg=cfg).cpu().double()
    ref16 = A.half().double() @ W.half().double().t() + bias.double()
    ref = A.double() @ W.double().t() + bias.double()
    assert rel_l2(got.numpy(), ref16.numpy()) < 2e-6
    assert rel_l2(got.numpy(), ref.numpy()) < 6e-4
    # saturating conversion: finite for values beyond the fp16 range, subnormal inputs are not flushed
    A2 = A.clone(); A2[0, :] = 1e6; A2[1, :] = 3e-7
    got2 = eng.test_gemm3(A2, W, None, split=2, cfg=cfg).cpu().double()
    assert torch.isfinite(got2).all()
    r1 = (A2[1].half().double()[None] @ W.half().double().t())[0]
    assert rel_l2(got2[1].numpy(), r1.numpy()) < 1e-5 and float(r1.abs().max()) > 0


def test_default_precision_teacher_128_steps_stays_inside_the_contract(eng, dit_weights):
    """BASELINE config 5 at the SHIPPED precision: 128 chained ODE + CFG steps (distill.py:60-134 ingredients) against the fp32
    oracle, x0-hat of every step — fp16 operand rounding is re-injected 128 times, so this is where drift would show.  The
    split-bf16 version of this test (test_dit_gpu.py) asserts 5e-4; here the north star's 1e-3 contract itself is the bar."""
    gen = torch.Generator().manual_seed(51)
    B, N, R, P, steps = 2, 20, 6, 9, 128
    ref = torch.randn(B, R, 64, generator=gen)
    ids = torch.randint(1, 198, (B, P), generator=gen)
    pm = torch.ones(B, P, dtype=torch.bool); rl = torch.full((B,), R)
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, 15:] = False
    noise = torch.randn(B, N, 64, generator=gen)
    ref3, len3, ids3, pm3 = O.cfg_conditions(ref, rl, ids, pm)
    keep = []
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, ref3, len3, ids3, pm3)
        ox = O.sample_teacher_ode(dit_weights, oc, pm3, mask, noise, steps, keep=keep)
    cache3 = eng.cond_encode(ref3, len3, ids3, pm3)
    x, per_step = eng.sample(cache3, mask, num_steps=steps, mode="ode", cfg=True, noise=noise, return_steps=True)
    m = mask.numpy()
    errs = [rel_l2(per_step[i].cpu().numpy()[m], keep[i].numpy()[m]) for i in range(steps)]
    end = rel_l2(x.cpu().numpy()[m], ox.numpy()[m])
    print(f"\n[teacher 128 @ f16 mixed] x0-hat rel L2 vs oracle: step 0 {errs[0]:.2e}, 31 {errs[31]:.2e}, 63 {errs[63]:.2e}, "
          f"127 {errs[127]:.2e}, max {max(errs):.2e} at step {int(np.argmax(errs))}; final latents {end:.2e}")
    assert max(errs) < 1e-3 and end < 1e-3, f"128-step chain at the default precision: max {max(errs):.3e}, end {end:.3e}"


def test_default_precision_teacher_cfg_at_bench_size_vs_oracle(eng, dit_weights):
    """Config 5's row count at the shipped precision: B = 8 x N = 75 with CFG = 1800 rows per denoiser call — the shapes that take
    the one-round tile choices of the single-array formats (128x128 QKVG on shallow rings, 160x128 SwiGLU pairs), the resident-K/V
    attention form (24 x 8 workgroups) and the per-group conv pos-embed over 1800 rows; two chained ODE + CFG steps."""
    gen = torch.Generator().manual_seed(15)
    B, N, R, P, steps = 8, 75, 15, 30, 2
    ref = torch.randn(B, R, 64, generator=gen)
    ids = torch.randint(1, 198, (B, P), generator=gen)
    pm = torch.ones(B, P, dtype=torch.bool); rl = torch.full((B,), R)
    mask = torch.ones(B, N, dtype=torch.bool)
    noise = torch.randn(B, N, 64, generator=gen)
    ref3, len3, ids3, pm3 = O.cfg_conditions(ref, rl, ids, pm)
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, ref3, len3, ids3, pm3)
        ox = O.sample_teacher_ode(dit_weights, oc, pm3, mask, noise, steps)
    cache3 = eng.cond_encode(ref3, len3, ids3, pm3)
    x = eng.sample(cache3, mask, num_steps=steps, mode="ode", cfg=True, noise=noise).cpu().numpy()
    err = rel_l2(x, ox.numpy())
    print(f"\n[teacher CFG, 1800 rows @ f16 mixed] latents rel L2 vs oracle {err:.2e}")
    assert err < 1e-3, f"teacher ODE at 1800 rows, default precision: rel L2 {err:.3e}"


# ---- the configuration bench.py TIMES: throughput tuning, three batches in flight, default precision,
# ---- bench.make_inputs shapes — against the oracle, not against itself -------------------------------------------------------
def _three_in_flight(eng, fn):
    """Runs fn(i) for i = 0..2, each whole on its own HIP stream with its own workspace (bench.run_steps' issue pattern)."""
    dev = eng.device
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)
    outs = [None] * 3
    try:
        for i in range(3):
            with torch.cuda.stream(streams[i]):
                eng.use_workspace(f"batch{i}")
                outs[i] = fn(i)
    finally:
        eng.use_workspace(None)
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    return outs


@pytest.fixture(scope="module")
def bench_oracle(dit_weights, golden_seed):
    """Oracle latents (fp32 torch restatement of the reference modules, pinned by tests/golden) for bench.make_inputs' batch
    with injected sampler noise, and the oracle's decode of two of its eight utterances."""
    import bench
    inp = bench.make_inputs(torch.device("cpu"), 0)
    noise = torch.randn(4, bench.B, bench.N_FRAMES, 64, generator=torch.Generator().manual_seed(1234))
    wd = O.to_torch(synth_state_dict(codec_decoder_param_specs(DEFAULT_CODEC), golden_seed))
    with torch.no_grad():
        oc = O.encode_conditions(dit_weights, inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
        ox = O.sample_dmd(dit_weights, oc, inp["ph_mask"], inp["mask"], noise, 4)
        audio = {b: CO.decode(wd, ox[b:b + 1], DEFAULT_CODEC).numpy() for b in (1, 6)}
    return inp, noise, ox.numpy(), audio, wd


def test_timed_configuration_latents_vs_oracle_alone_and_in_flight(eng, bench_oracle):
    """cond_encode + 4-step sample under THROUGHPUT tuning (unsplit M = 600 GEMMs with the EpiResid<1> epilogue + separate
    ln_modulate, no engine side stream) at B=8, N=75, R=15, P=30, default precision: < 3e-4 of the oracle alone and as each of
    three batches in flight on three streams / workspaces (onnx.py:91-125)."""
    inp, noise, ox, _, _ = bench_oracle

    def latents(_i=0):
        cache = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
        return eng.sample(cache, inp["mask"], num_steps=4, noise=noise)

    lat_latency = latents().cpu().numpy()
    prev = eng.set_tuning("throughput")
    try:
        alone = latents().cpu().numpy()
        flight = [x.cpu().numpy() for x in _three_in_flight(eng, latents)]
    finally:
        eng.set_tuning(prev)
    e_lat, e_alone = rel_l2(lat_latency, ox), rel_l2(alone, ox)
    e_fl = [rel_l2(x, ox) for x in flight]
    print(f"\n[timed configuration] latent rel-L2 vs oracle: latency tuning {e_lat:.2e}; throughput tuning alone {e_alone:.2e}, "
          f"three in flight {', '.join(f'{e:.2e}' for e in e_fl)}")
    assert e_alone < TOL_F16, f"throughput tuning alone: {e_alone:.3e}"
    for i, e in enumerate(e_fl):
        assert e < TOL_F16, f"throughput tuning, batch {i} of three in flight: {e:.3e}"
        assert np.array_equal(flight[i], alone), f"batch {i} in flight differs from the same batch alone"


def test_timed_configuration_codec_decode_vs_oracle(eng, bench_oracle):
    """Full-spec codec_decode of the 8 x 75 bench batch under throughput tuning (persistent kernels on 192 CUs), alone and three
    in flight: two utterances against the CPU codec oracle > 64 dB (codec/onnx.py:42-53; oracle parity-unpinned)."""
    _, _, ox, audio, _ = bench_oracle
    lat = torch.from_numpy(ox).to(eng.device)
    base = eng.codec_decode(lat).cpu().numpy()
    prev = eng.set_tuning("throughput")
    try:
        alone = eng.codec_decode(lat).cpu().numpy()
        flight = [x.cpu().numpy() for x in _three_in_flight(eng, lambda i: eng.codec_decode(lat))]
    finally:
        eng.set_tuning(prev)
    for b, r in audio.items():
        s = snr_db(alone[b:b + 1], r)
        print(f"\n[timed configuration] decode, utterance {b}: {s:.1f} dB vs oracle under throughput tuning")
        assert s > SNR_F16, f"utterance {b}: decode SNR {s:.1f} dB under throughput tuning"
    assert snr_db(alone, base) > 80.0
    for i, x in enumerate(flight):
        assert np.array_equal(x, alone), f"decode {i} of three in flight differs from the same decode alone"


def test_timed_configuration_end_to_end_audio_vs_oracle(eng, bench_oracle):
    """The whole call the reference makes (onnx.py:68-129): HIP cond_encode -> HIP 4-step sample -> HIP codec_decode, exactly as
    bench.one_step issues it (throughput tuning, three batches in flight, default precision, full-size 344 M-parameter decoder),
    against oracle sample -> oracle decode on 2 of 8 utterances.  Bound stated by this build for the COMPOSED path: 60 dB,
    asserted at 64 (measured 68.3 dB: the latent error of 1.3e-4 = 78 dB adds almost nothing to the decoder's own 68.9 dB)."""
    inp, noise, ox, audio, _ = bench_oracle

    def step(_i):
        cache = eng.cond_encode(inp["ref"], inp["ref_len"], inp["ids"], inp["ph_mask"])
        return eng.codec_decode(eng.sample(cache, inp["mask"], num_steps=4, noise=noise))

    prev = eng.set_tuning("throughput")
    try:
        outs = [x.cpu().numpy() for x in _three_in_flight(eng, step)]
    finally:
        eng.set_tuning(prev)
    assert outs[0].shape == (8, 1, 3200 * 75)
    for i in (1, 2):
        assert np.array_equal(outs[i], outs[0])
    for b, r in audio.items():
        s = snr_db(outs[0][b:b + 1], r)
        print(f"\n[timed configuration] end to end, utterance {b}: audio SNR {s:.1f} dB vs oracle sample -> oracle decode")
        assert s > E2E_SNR, f"utterance {b}: end-to-end audio SNR {s:.1f} dB"
