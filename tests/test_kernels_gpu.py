"""GPU: single kernels through the C ABI vs plain torch fp32/fp64 on the same seeded inputs."""
import numpy as np
import pytest
import torch

from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from smalltts_amd.engine import HipEngine
    return HipEngine(0, "bf16x3")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_synth_matches_numpy_bitwise(eng):
    from smalltts_amd.weights import init_rule, synth_tensor, tensor_key
    import ctypes as C
    for name, shape in (("dit.transformer_blocks.3.ff.w1.weight", (2400, 960)), ("velocity.bias", (64,)),
                        ("dit.transformer_blocks.0.attn.q_norm.weight", (8, 120)), ("style_encoder.log_scale", ())):
        mean, hr = init_rule(name, shape)
        sh = (C.c_int64 * max(1, len(shape)))(*shape)
        assert eng.lib.smtts_synth_tensor(eng.h, name.encode(), sh, len(shape), C.c_uint64(tensor_key(name, 5)), mean, hr) == 0
        got = eng.get_tensor(name, shape)
        assert np.array_equal(got, synth_tensor(name, shape, 5)), name


# (M, N, K): hot-path shapes + ragged edges; asymmetric data catches transposed fragments
GEMM_SHAPES = [(600, 3840, 960), (600, 960, 2400), (600, 64, 960), (120, 2048, 512), (75, 60, 1984),
               (4, 960, 256), (37, 100, 96), (1000, 32, 128), (5000, 128, 32), (300, 8192, 2048), (129, 130, 40)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_split3_fp32_class(eng, M, N, K):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2) / K ** 0.5, _rand(N, seed=3)
    ref = (A.double() @ W.double().t() + b.double())
    got = eng.test_gemm(A, W, b, split=3).cpu()
    err = rel_l2(got.numpy(), ref.numpy())
    assert err < 2e-5, f"split-bf16 gemm {M}x{N}x{K}: rel l2 {err:.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4])
def test_gemm_every_tile_config(eng, cfg):
    M, N, K = 333, 200, 160
    A, W = _rand(M, K, seed=4), _rand(N, K, seed=5) / K ** 0.5
    ref = A.double() @ W.double().t()
    for split, tol in ((3, 2e-5), (1, 1e-2)):
        got = eng.test_gemm(A, W, None, split=split, cfg=cfg).cpu()
        err = rel_l2(got.numpy(), ref.numpy())
        assert err < tol, f"cfg {cfg} split {split}: {err:.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("K", [64, 128, 192, 960])
def test_gemm_v1_every_config_long_and_short_k(eng, cfg, K):
    """The fp32-A (register-staged) kernel of the cold paths: nk = 1, 2, 3, 15 k-tiles, with bias."""
    M, N = 333, 200
    A, W, b = _rand(M, K, seed=40 + cfg), _rand(N, K, seed=41) / K ** 0.5, _rand(N, seed=42)
    ref = A.double() @ W.double().t() + b.double()
    for split, tol in ((3, 2e-5), (1, 1e-2)):
        err = rel_l2(eng.test_gemm(A, W, b, split=split, cfg=cfg).cpu().numpy(), ref.numpy())
        assert err < tol, f"v1 cfg {cfg} K {K} split {split}: {err:.3e}"


def test_gemm_v1_repeatable(eng):
    A, W = _rand(600, 960, seed=50), _rand(3840, 960, seed=51) / 31.0
    outs = [eng.test_gemm(A, W, None, split=3).cpu() for _ in range(4)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])


G3_SHAPES = [(600, 3840, 960), (600, 960, 2432), (600, 64, 960), (120, 2048, 512), (75, 60, 1984), (37, 100, 64),
             (1000, 32, 128), (5000, 128, 64), (300, 8192, 2048), (129, 130, 192), (240, 23040, 960),
             (600, 8192, 512), (4800, 1024, 256), (601, 8200, 128)]  # the last three pick the 160x128 one-round tile


@pytest.mark.parametrize("M,N,K", G3_SHAPES)
def test_gemm3_hot_path_kernel(eng, M, N, K):
    """gemm3 (split-bf16 A and W through the DMA ring, 8 waves): fp32-class vs fp64, auto tile choice."""
    A, W, b = _rand(M, K, seed=60), _rand(N, K, seed=61) / K ** 0.5, _rand(N, seed=62)
    ref = A.double() @ W.double().t() + b.double()
    err = rel_l2(eng.test_gemm3(A, W, b, split=3).cpu().numpy(), ref.numpy())
    assert err < 2e-5, f"gemm3 {M}x{N}x{K}: {err:.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])   # 5 = 160x128 (10 waves, surplus DMA slots)
@pytest.mark.parametrize("K", [64, 128, 192, 960])
def test_gemm3_every_config_and_short_k(eng, cfg, K):
    M, N = 333, 200
    A, W, b = _rand(M, K, seed=70 + cfg), _rand(N, K, seed=71) / K ** 0.5, _rand(N, seed=72)
    ref = A.double() @ W.double().t() + b.double()
    for split, tol in ((3, 2e-5), (1, 1e-2)):
        err = rel_l2(eng.test_gemm3(A, W, b, split=split, cfg=cfg).cpu().numpy(), ref.numpy())
        assert err < tol, f"gemm3 cfg {cfg} K {K} split {split}: {err:.3e}"
    gelu = torch.nn.functional.gelu(ref.float()).numpy()
    err = rel_l2(eng.test_gemm3(A, W, b, act="gelu", cfg=cfg).cpu().numpy(), gelu)
    assert err < 3e-5, f"gemm3 gelu cfg {cfg}: {err:.3e}"


def test_gemm3_repeatable(eng):
    A, W = _rand(600, 960, seed=80), _rand(3840, 960, seed=81) / 31.0
    outs = [eng.test_gemm3(A, W, None, split=3).cpu() for _ in range(6)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_gemm_bf16_single_pass_error_level(eng):
    A, W = _rand(600, 960, seed=6), _rand(960, 960, seed=7) / 960 ** 0.5
    ref = A.double() @ W.double().t()
    err = rel_l2(eng.test_gemm(A, W, None, split=1).cpu().numpy(), ref.numpy())
    assert 1e-4 < err < 6e-3, f"plain bf16 error {err:.3e} outside the expected bf16 band"


@pytest.mark.parametrize("act", ["silu", "gelu", "mish"])
def test_gemm_activations(eng, act):
    A, W, b = _rand(70, 96, seed=8), _rand(130, 96, seed=9) / 96 ** 0.5, _rand(130, seed=10)
    z = A @ W.t() + b
    ref = {"silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu, "mish": torch.nn.functional.mish}[act](z)
    err = rel_l2(eng.test_gemm(A, W, b, act=act).cpu().numpy(), ref.numpy())
    assert err < 3e-5, f"{act}: {err:.3e}"


@pytest.mark.parametrize("M,F,K", [(600, 2400, 960), (120, 1536, 512), (50, 64, 64)])
def test_swiglu(eng, M, F, K):
    A = _rand(M, K, seed=11)
    W1, W3 = _rand(F, K, seed=12) / K ** 0.5, _rand(F, K, seed=13) / K ** 0.5
    b1, b3 = _rand(F, seed=14), _rand(F, seed=15)
    ref = torch.nn.functional.silu(A.double() @ W1.double().t() + b1.double()) * (A.double() @ W3.double().t() + b3.double())
    err = rel_l2(eng.test_swiglu(A, W1, W3, b1, b3).cpu().numpy(), ref.numpy())
    assert err < 3e-5, f"swiglu: {err:.3e}"
    ref0 = torch.nn.functional.silu(A.double() @ W1.double().t()) * (A.double() @ W3.double().t())
    err = rel_l2(eng.test_swiglu(A, W1, W3).cpu().numpy(), ref0.numpy())
    assert err < 3e-5, f"swiglu no-bias: {err:.3e}"


def _attn_ref(qkvg, qw, kw, eps, rope, rot, H, dh, kr, vr, kt, vt, ms, mr, mt):
    """torch restatement of dit.py:95-119 on raw projections (fp64)."""
    B, N, _ = qkvg.shape
    D = H * dh
    x = qkvg.double()
    q, k, v, g = (x[..., i * D:(i + 1) * D].reshape(B, N, H, dh) for i in range(4))
    rms = lambda t, w: t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + eps) * w.double()
    q, k = rms(q, qw), rms(k, kw)

    def rot_pairs(t):
        a = rope.double()[:N, :rot][None, :, None, 0::2]
        te, to = t[..., 0:rot:2], t[..., 1:rot:2]
        out = t.clone()
        out[..., 0:rot:2] = te * a.cos() - to * a.sin()
        out[..., 1:rot:2] = to * a.cos() + te * a.sin()
        return out
    q, k = rot_pairs(q).transpose(1, 2), rot_pairs(k).transpose(1, 2)
    v = v.transpose(1, 2)
    keys, vals, masks = [k], [v], [ms if ms is not None else torch.ones(B, N, dtype=torch.bool)]
    for kk, vv, mm in ((kr, vr, mr), (kt, vt, mt)):
        if kk is not None:
            keys.append(kk.double()); vals.append(vv.double())
            masks.append(mm if mm is not None else torch.ones(B, kk.shape[2], dtype=torch.bool))
    K, V, Mk = torch.cat(keys, 2), torch.cat(vals, 2), torch.cat(masks, 1)
    s = q @ K.transpose(-1, -2) / dh ** 0.5
    s = s.masked_fill(~Mk[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ V).transpose(1, 2).reshape(B, N, D)
    return o * torch.sigmoid(g.reshape(B, N, D))


@pytest.mark.parametrize("B,N,H,dh,rot,R,P", [(2, 75, 8, 120, 64, 15, 30), (3, 21, 8, 64, 64, 0, 0),
                                             (2, 40, 4, 128, 128, 0, 0), (2, 130, 8, 120, 64, 70, 90),
                                             (1, 5, 8, 120, 64, 3, 2), (16, 75, 8, 120, 64, 15, 30)])   # last: resident-K/V form
@pytest.mark.parametrize("mfma", [False, "img:bf16x3", "img:f16", "img:bf16"])
def test_attention(eng, B, N, H, dh, rot, R, P, mfma):
    D = H * dh
    qkvg = _rand(B, N, 4 * D, seed=20)
    qw, kw = 1 + 0.2 * _rand(H, dh, seed=21), 1 + 0.2 * _rand(H, dh, seed=22)
    inv = 1.0 / (1e4 ** (torch.arange(0, rot, 2).float() / rot))
    rope = (torch.arange(max(N, 1)).float()[:, None] * inv[None]).repeat_interleave(2, -1).contiguous()
    ms = torch.ones(B, N, dtype=torch.bool); ms[-1, N - N // 4:] = False
    kr = vr = kt = vt = mr = mt = None
    if R:
        kr, vr = _rand(B, H, R, dh, seed=23), _rand(B, H, R, dh, seed=24)
        mr = torch.ones(B, R, dtype=torch.bool); mr[0, R // 2:] = False
    if P:
        kt, vt = _rand(B, H, P, dh, seed=25), _rand(B, H, P, dh, seed=26)
        mt = torch.ones(B, P, dtype=torch.bool); mt[-1, :] = False  # a fully masked segment
    ref = _attn_ref(qkvg, qw, kw, 1e-6, rope, rot, H, dh, kr, vr, kt, vt, ms, mr, mt)
    got = eng.test_attention(qkvg, qw, kw, 1e-6, rope, rot, H, dh, kr, vr, kt, vt, ms, mr, mt, mfma=mfma).cpu()
    err = rel_l2(got.numpy(), ref.numpy())
    tol = {"img:f16": 1.5e-3, "img:bf16": 1.2e-2}.get(mfma, 3e-5 if mfma else 1e-5)   # single-pass operand formats: 2^-11 / 2^-8 rounding
    assert err < tol, f"attention (mfma={mfma}): {err:.3e}"


def test_attention_many_tiles_per_workgroup_and_streamed_chunks_agree_with_the_reference(eng):
    """Grids above two workgroups per CU make a workgroup walk several query tiles against keys staged once (B x H = 512 here); more
    key chunks than LDS slots stream through two slots (Ktot = 700 at split-bf16: 11 chunks).  Both against the fp64 reference."""
    for (B, N, H, dh, rot, R, P, mode, tol) in [(64, 75, 8, 120, 64, 15, 30, "img:bf16x3", 3e-5), (64, 75, 8, 120, 64, 15, 30, "img:f16", 1.5e-3),
                                                (1, 300, 8, 120, 64, 150, 250, "img:bf16x3", 3e-5), (2, 200, 4, 128, 128, 0, 0, "img:f16", 1.5e-3)]:
        D = H * dh
        qkvg = _rand(B, N, 4 * D, seed=40)
        qw, kw = 1 + 0.2 * _rand(H, dh, seed=41), 1 + 0.2 * _rand(H, dh, seed=42)
        inv = 1.0 / (1e4 ** (torch.arange(0, rot, 2).float() / rot))
        rope = (torch.arange(N).float()[:, None] * inv[None]).repeat_interleave(2, -1).contiguous()
        ms = torch.ones(B, N, dtype=torch.bool); ms[-1, N - 7:] = False
        kr = vr = kt = vt = mr = mt = None
        if R:
            kr, vr = _rand(B, H, R, dh, seed=43), _rand(B, H, R, dh, seed=44)
            kt, vt = _rand(B, H, P, dh, seed=45), _rand(B, H, P, dh, seed=46)
            mr = torch.ones(B, R, dtype=torch.bool); mr[0, 7:] = False
            mt = torch.ones(B, P, dtype=torch.bool); mt[-1, :] = False
        ref = _attn_ref(qkvg, qw, kw, 1e-6, rope, rot, H, dh, kr, vr, kt, vt, ms, mr, mt)
        got = eng.test_attention(qkvg, qw, kw, 1e-6, rope, rot, H, dh, kr, vr, kt, vt, ms, mr, mt, mfma=mode).cpu()
        err = rel_l2(got.numpy(), ref.numpy())
        assert err < tol, f"B={B} N={N} Ktot={N + R + P} {mode}: {err:.3e}"


def test_attention_all_keys_masked_gives_zero(eng):
    B, N, H, dh = 2, 9, 8, 64
    qkvg = _rand(B, N, 4 * H * dh, seed=30)
    w = torch.ones(H, dh)
    rope = torch.zeros(N, dh)
    ms = torch.ones(B, N, dtype=torch.bool); ms[1] = False
    for mfma in (False, "img:bf16x3", "img:f16"):
        got = eng.test_attention(qkvg, w, w, 1e-5, rope, dh, H, dh, mask_self=ms, mfma=mfma).cpu()
        assert torch.isfinite(got).all() and float(got[1].abs().max()) == 0.0 and float(got[0].abs().max()) > 0


def test_randn_matches_oracle_philox(eng):
    from oracle.philox import philox_randn
    got = eng.randn(4099, seed=1234, stream_id=7).cpu().numpy()
    ref = philox_randn(4099, 1234, 7)
    assert np.abs(got - ref).max() < 2e-5
    big = eng.randn(1 << 20, seed=1, stream_id=0).cpu().numpy()
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1) < 5e-3


@pytest.mark.parametrize("sr,target", [(16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000)])
def test_device_resampler_matches_oracle(eng, sr, target):
    """the device polyphase resampler against oracle/resample_oracle.py (float64, closed form per sample pair:
    no bank, no framing — nothing shared with the product's bank builder)."""
    from oracle.resample_oracle import resample as oracle_resample
    rng = np.random.default_rng(sr)
    t = np.arange(int(0.2 * sr)) / sr
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    want = oracle_resample(x, sr, target)
    got = eng.resample(x, sr, target).cpu().numpy()
    assert got.shape == want.shape
    assert rel_l2(got, want) < 3e-6
    two = eng.resample(np.stack([x, -x]), sr, target).cpu().numpy()
    assert two.shape == (2, want.size) and rel_l2(two[1], -want) < 3e-6
    assert torch.equal(eng.resample(x, target, target).cpu(), torch.from_numpy(x))


def test_device_pcm16_is_bit_exact_with_the_wav_writer(eng, tmp_path):
    from smalltts_amd.audio import read_wav, write_wav_pcm16
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1.3, 1.3, 5000), [0.0, 1.0, -1.0, 0.5 / 32767, 1.5 / 32767, -2.5 / 32767]]).astype(np.float32)
    write_wav_pcm16(str(tmp_path / "a.wav"), x)
    host = np.frombuffer(open(tmp_path / "a.wav", "rb").read()[44:], "<i2")
    dev = eng.pcm16(x).cpu().numpy()
    assert dev.dtype == np.int16 and np.array_equal(dev, host)


@pytest.mark.parametrize("prec", ["f16", "bf16x3"])
def test_qkvg_epilogue_writes_the_images_the_pack_kernel_writes_bit_for_bit(prec, monkeypatch):
    """gemm3 EpiQKV (head RMSNorm + RoPE + operand formatting on the GEMM's accumulators, through an LDS tile) against the
    stand-alone qkv_pack kernel on the fp32 projection of the same GEMM: the same arithmetic in the same order, so the whole
    chain behind them — encoders (dh 64 / 128, unpadded heads), DiT blocks (dh 120 padded to 128 columns), ragged masks,
    cross keys — must agree bit for bit.  Also at M = 1800 rows (the teacher's 128x128 tiles)."""
    from smalltts_amd.engine import HipEngine
    engs = {}
    for epi in ("1", "0"):
        monkeypatch.setenv("SMTTS_ATTN_EPI", epi)
        e = HipEngine(0, prec); e.load_synthetic(7, parts=("dit",)); e.finalize()
        e.set_ln_fold(False)   # (the norm fold lives in the GEMM epilogues: the pack-kernel path has none, so both run the norm launches)
        engs[epi] = e
    for (B, N, R, P) in [(3, 75, 15, 30), (2, 33, 9, 70), (24, 75, 15, 30)]:
        g = torch.Generator().manual_seed(B * 100 + N)
        ref = torch.randn(B, R, 64, generator=g); ids = torch.randint(1, 198, (B, P), generator=g)
        rl = torch.full((B,), R); rl[-1] = max(1, R // 2)
        pm = torch.ones(B, P, dtype=torch.bool); pm[0, P // 3:] = False
        mask = torch.ones(B, N, dtype=torch.bool); mask[-1, N - 5:] = False
        xt = torch.randn(B, N, 64, generator=g); t = torch.full((B,), 0.6)
        outs = {}
        for epi, e in engs.items():
            c = e.cond_encode(ref, rl, ids, pm, debug=True)
            v = e.denoise_step(xt, mask, t, c)
            outs[epi] = {k: x.clone() for k, x in c.items() if torch.is_tensor(x)} | {"velocity": v.clone()}
        for k in outs["1"]:
            assert torch.equal(outs["1"][k], outs["0"][k]), f"{prec} B={B} N={N}: {k} differs, max {float((outs['1'][k].float() - outs['0'][k].float()).abs().max()):.3e}"
    for e in engs.values():
        e.close()
