// extern "C" surface of libsmalltts_hip.so — see include/smalltts_hip.h for the contract.
#include "../../include/smalltts_hip.h"

#include <cstring>
#include <string>

#include "engine.hpp"

struct smtts_engine {
    Engine e;
    explicit smtts_engine(int dev) : e(dev) {}
};
static thread_local std::string g_create_err;

#define E (h->e)
// every entry point that takes a handle rejects NULL with an error code (message via smtts_last_error(NULL))
#define NULLCHK if (!h) { g_create_err = std::string(__func__) + ": null handle"; return 1; }
#define NULLCHK0 if (!h) { g_create_err = std::string(__func__) + ": null handle"; return 0; }
#define ST(s) static_cast<hipStream_t>(s)

extern "C" {

const char* smtts_version(void) { return "smalltts-hip 0.5 (gfx950)"; }
int smtts_abi_version(void) { return SMTTS_ABI_VERSION; }

int smtts_create(int device_id, smtts_handle* out) {
    if (!out) return 1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device_id < 0 || device_id >= n) {
        g_create_err = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e)
                                       : "device id out of range (have " + std::to_string(n) + " GPUs)";
        return 1;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); return 1; }
    *out = new smtts_engine(device_id);
    return 0;
}
int smtts_destroy(smtts_handle h) { delete h; return 0; }  // NULL is a no-op, like free()
const char* smtts_last_error(smtts_handle h) { return h ? E.last_error().c_str() : g_create_err.c_str(); }

int smtts_set_tensor(smtts_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_dev) { NULLCHK;
    long sh[8];
    if (ndim > 8) return E.fail("set_tensor: ndim > 8");
    for (int i = 0; i < ndim; ++i) sh[i] = (long)shape[i];
    return E.set_tensor(name, data, sh, ndim, on_dev != 0);
}
int smtts_synth_tensor(smtts_handle h, const char* name, const int64_t* shape, int ndim, uint64_t key, float mean,
                       float half_range) { NULLCHK;
    long sh[8];
    if (ndim > 8) return E.fail("synth_tensor: ndim > 8");
    for (int i = 0; i < ndim; ++i) sh[i] = (long)shape[i];
    return E.synth_tensor(name, sh, ndim, key, mean, half_range);
}
int smtts_get_tensor(smtts_handle h, const char* name, float* host_out, int64_t numel) { NULLCHK;
    return E.get_tensor(name, host_out, (long)numel);
}
int smtts_set_codec_spec(smtts_handle h, int latent_dim, int n_filters, int kernel, int ffn_mult, float eps,
                         const int* ratios, int n_ratios, const int* depths) { NULLCHK;
    CodecSpecC s;
    if (n_ratios < 1 || n_ratios > 7) return E.fail("codec spec: 1..7 ratios supported");
    s.latent_dim = latent_dim; s.n_filters = n_filters; s.kernel = kernel; s.ffn_mult = ffn_mult; s.eps = eps;
    s.n_ratios = n_ratios;
    for (int i = 0; i < 8; ++i) s.ratios[i] = i < n_ratios ? ratios[i] : 0;
    for (int i = 0; i < 9; ++i) s.depths[i] = i <= n_ratios ? depths[i] : 0;
    return E.set_codec_spec(s);
}
int smtts_finalize(smtts_handle h) { NULLCHK; return E.finalize(); }
int smtts_set_precision(smtts_handle h, int preset) { NULLCHK;
    if (preset < 1 || preset > 3) return E.fail("precision preset must be 1 (bf16), 2 (f16 mixed) or 3 (split-bf16)");
    E.set_precision(preset);
    return 0;
}
int smtts_get_precision(smtts_handle h) { NULLCHK; return E.precision(); }
int smtts_default_precision(void) { return kDefaultPrecision; }
int smtts_set_site_precision(smtts_handle h, int site, int prec) { NULLCHK; return E.set_site_precision(site, prec); }
int smtts_get_saturations(smtts_handle h, uint32_t* counts, int n_sites, int reset) { NULLCHK;
    if (!counts || n_sites < 0) return E.fail("get_saturations: bad arguments");
    return E.get_saturations(counts, n_sites, reset != 0);
}
const char* smtts_range_report(smtts_handle h) {
    if (!h) { g_create_err = std::string(__func__) + ": null handle"; return ""; }
    return E.range_report().c_str();
}
float smtts_range_worst_bound(smtts_handle h) { NULLCHK0; return E.range_worst(); }
int smtts_has_part(smtts_handle h, int part) { NULLCHK0;
    return part == 0 ? E.has_dit() : part == 1 ? E.has_decoder() : part == 2 ? E.has_encoder() : 0;
}

size_t smtts_cond_workspace_bytes(smtts_handle h, int B, int R, int P) { NULLCHK0; return E.cond_ws_bytes(B, R, P); }
int smtts_cond_encode(smtts_handle h, void* stream, const float* ref, const int64_t* ref_len, const int64_t* phonemes,
                      const uint8_t* ph_mask, int B, int R, int P, float* k_ref, float* v_ref, uint8_t* ref_mask,
                      float* k_text, float* v_text, void* ws, size_t ws_bytes, float* ref_seq_out, float* mem_out) { NULLCHK;
    return E.cond_encode(ST(stream), ref, ref_len, phonemes, ph_mask, B, R, P, k_ref, v_ref, ref_mask, k_text, v_text, ws,
                         ws_bytes, ref_seq_out, mem_out);
}

size_t smtts_denoise_workspace_bytes(smtts_handle h, int B, int N, int R, int P) { NULLCHK0; return E.denoise_ws_bytes(B, N, R, P, B); }
int smtts_denoise_step(smtts_handle h, void* stream, const float* x_t, const uint8_t* mask, const float* t,
                       const float* k_ref, const float* v_ref, const uint8_t* ref_mask, const float* k_text,
                       const float* v_text, const uint8_t* ph_mask, const float* rope, int B, int N, int R, int P,
                       float* velocity, void* ws, size_t ws_bytes) { NULLCHK;
    return E.denoise_step(ST(stream), x_t, mask, t, k_ref, v_ref, ref_mask, k_text, v_text, ph_mask, rope, B, N, R, P,
                          velocity, ws, ws_bytes);
}

size_t smtts_sample_workspace_bytes(smtts_handle h, int B, int N, int R, int P, int n_steps, int cfg) { NULLCHK0;
    return E.sample_ws_bytes(B, N, R, P, n_steps, cfg);
}
int smtts_sample(smtts_handle h, void* stream, int mode, int n_steps, int cfg, float s_text, float s_spk,
                 const uint8_t* mask, const float* k_ref, const float* v_ref, const uint8_t* ref_mask,
                 const float* k_text, const float* v_text, const uint8_t* ph_mask, int B, int N, int R, int P,
                 const float* noise, uint64_t seed, float* x_out, float* steps_out, void* ws, size_t ws_bytes) { NULLCHK;
    return E.sample(ST(stream), mode, n_steps, cfg, s_text, s_spk, mask, k_ref, v_ref, ref_mask, k_text, v_text, ph_mask,
                    B, N, R, P, noise, seed, x_out, steps_out, ws, ws_bytes);
}

int smtts_codec_hop(smtts_handle h) { NULLCHK0; return E.codec_spec().hop(); }
size_t smtts_decode_workspace_bytes(smtts_handle h, int B, int T) { NULLCHK0; return E.decode_ws_bytes(B, T); }
int smtts_codec_decode(smtts_handle h, void* stream, const float* latents, int B, int T, float* audio, void* ws,
                       size_t ws_bytes) { NULLCHK;
    return E.codec_decode(ST(stream), latents, B, T, audio, ws, ws_bytes);
}
size_t smtts_encode_workspace_bytes(smtts_handle h, int B, int S) { NULLCHK0; return E.encode_ws_bytes(B, S); }
int smtts_codec_encode(smtts_handle h, void* stream, const float* audio, int B, int S, float* latents, void* ws,
                       size_t ws_bytes) { NULLCHK;
    return E.codec_encode(ST(stream), audio, B, S, latents, ws, ws_bytes);
}

int smtts_randn(smtts_handle h, void* stream, float* out, int64_t n, uint64_t seed, uint64_t stream_id) { NULLCHK;
    hipError_t e = launch_randn(out, (long)n, seed, stream_id, ST(stream));
    return e == hipSuccess ? 0 : E.fail_hip(e, "randn");
}

void smtts_alpha_sigma(float t, float* alpha, float* sigma) { alpha_sigma_host(t, *alpha, *sigma); }

int smtts_resample_poly(smtts_handle h, void* stream, const float* x, int channels, int64_t n_in, const float* bank, int up,
                        int down, int klen, int width, float* y, int64_t n_out) { NULLCHK;
    if (up <= 0 || down <= 0 || klen <= 0 || channels <= 0) return E.fail("resample_poly: bad arguments");
    hipError_t e = launch_resample_poly(x, n_in, bank, up, down, klen, width, y, n_out, channels, ST(stream));
    return e == hipSuccess ? 0 : E.fail_hip(e, "resample_poly");
}
int smtts_pcm16(smtts_handle h, void* stream, const float* x, int64_t n, int16_t* y) { NULLCHK;
    hipError_t e = launch_pcm16(x, y, n, ST(stream));
    return e == hipSuccess ? 0 : E.fail_hip(e, "pcm16");
}
int smtts_set_dual_stream(smtts_handle h, int on) { NULLCHK; E.set_dual_stream(on != 0); return 0; }
int smtts_set_tuning(smtts_handle h, int mode) { NULLCHK;
    if (mode != 0 && mode != 1) return E.fail("tuning mode must be 0 (latency) or 1 (throughput)");
    E.set_tuning(mode);
    return 0;
}
int smtts_profile_enable(smtts_handle h, int on) { NULLCHK; E.profile_enable(on); return 0; }
int smtts_profile_report(smtts_handle h, char* buf, size_t cap) { NULLCHK;
    std::string r = E.profile_report();
    if (r.size() + 1 > cap) return E.fail("profile_report: buffer too small");
    memcpy(buf, r.c_str(), r.size() + 1);
    return 0;
}

int smtts_bench_gemm(smtts_handle h, int M, int N, int K, int epi, int split, int cfg, int iters, int ver, float* avg_us) { NULLCHK;
    return E.bench_gemm(M, N, K, epi, split, cfg, iters, ver, avg_us);
}
int smtts_test_gemm3(smtts_handle h, void* stream, const float* A, const float* W, const float* bias, int M, int N, int K,
                     int act, int split, int cfg, float* C) { NULLCHK;
    return E.test_gemm3(ST(stream), A, W, bias, M, N, K, act, split, cfg, C);
}
int smtts_test_set_fused_ffn(smtts_handle h, int on) { NULLCHK; E.set_fused_ffn(on != 0); return 0; }
int smtts_test_set_ln_fold(smtts_handle h, int on) { NULLCHK; E.set_ln_fold(on != 0); return 0; }
int smtts_test_set_attention_mfma(smtts_handle h, int mode) { NULLCHK;   // 0: fp32 projection + qk_prep + the fp32 VALU reference kernel; else (default): producer-written operand images + the DMA / MFMA kernel
    E.set_attn_img(mode != 0);
    return 0;
}

// ---- test hooks -------------------------------------------------------------------------------
int smtts_test_gemm(smtts_handle h, void* stream, const float* A, int lda, const float* W, const float* bias, int M,
                    int N, int K, int act, int split, int cfg, float* C, int ldc) { NULLCHK;
    return E.test_gemm(ST(stream), A, lda, W, bias, M, N, K, act, split, cfg, C, ldc);
}
int smtts_test_swiglu(smtts_handle h, void* stream, const float* A, const float* W1, const float* W3, const float* b1,
                      const float* b3, int M, int F, int K, int split, float* out) { NULLCHK;
    return E.test_swiglu(ST(stream), A, W1, W3, b1, b3, M, F, K, split, out);
}
int smtts_test_attention(smtts_handle h, void* stream, const float* qkvg, const float* qw, const float* kw, float eps,
                         const float* rope, int rot_dim, const float* k_ref, const float* v_ref, int R,
                         const float* k_text, const float* v_text, int P, const uint8_t* mask_self,
                         const uint8_t* mask_ref, const uint8_t* mask_text, int B, int N, int H, int dh, float* out) { NULLCHK;
#ifndef SMTTS_TEST_KERNELS
    return E.fail("smtts_test_attention: the fp32 VALU reference attention is not part of this build (make TEST_KERNELS=1)");
#else
    AttnArgs a{};
    const int D = H * dh;
    a.q = qkvg; a.k = qkvg + D; a.v = qkvg + 2 * D; a.gate = qkvg + 3 * D;
    a.bs = (long)N * 4 * D; a.rs = 4 * D;
    float *rc = nullptr, *rs = nullptr;
    const int nr = N * rot_dim;
    if (hipMalloc(&rc, (size_t)nr * 4) != hipSuccess || hipMalloc(&rs, (size_t)nr * 4) != hipSuccess)
        return E.fail("test_attention: alloc failed");
    (void)launch_rope_cossin(rope, rc, rs, nr, ST(stream));
    a.qw = qw; a.kw = kw; a.eps = eps; a.rope_cos = rc; a.rope_sin = rs; a.rot_dim = rot_dim;
    a.k_ref = R > 0 ? k_ref : nullptr; a.v_ref = v_ref; a.R = R;
    a.k_text = P > 0 ? k_text : nullptr; a.v_text = v_text; a.P = P;
    a.mask_self = mask_self; a.mask_ref = mask_ref; a.mask_text = mask_text;
    a.out = out; a.obs = (long)N * D; a.ors = D;
    a.B = B; a.N = N; a.H = H; a.dh = dh;
    hipError_t e = launch_attention(a, ST(stream));
    (void)hipStreamSynchronize(ST(stream));
    (void)hipFree(rc);
    (void)hipFree(rs);
    return e == hipSuccess ? 0 : E.fail_hip(e, "attention");
#endif
}

int smtts_test_attention_mfma(smtts_handle h, void* stream, const float* qkvg, const float* qw, const float* kw, float eps,
                              const float* rope, int rot_dim, const float* k_ref, const float* v_ref, int R,
                              const float* k_text, const float* v_text, int P, const uint8_t* mask_self,
                              const uint8_t* mask_ref, const uint8_t* mask_text, int B, int N, int H, int dh, float* out) { NULLCHK;
    {
        // round-3 path: stand-alone producer (qkv_pack + cross_pack, L = 1) then the DMA + MFMA kernel, at the SITE_ATTN precision
        const int D = H * dh, dhp = dh <= 64 ? 64 : 128, Np = pad8(N), Rp = pad8(R > 0 ? R : 0), Cp = Rp + pad8(P > 0 ? P : 0);
        const int pa = E.site_precision(Engine::SITE_ATTN);
        const size_t nqk = (size_t)B * H * N * dhp, nvt = (size_t)B * H * dhp * Np, ng = (size_t)B * N * D, nc = (size_t)B * H * Cp * dhp;
        const int nr = N * rot_dim;
        bf16_t* img = nullptr; float *rc = nullptr, *rs = nullptr; bf16_t* ob = nullptr;
        const size_t tot = 2 * (2 * nqk + nvt + ng + 2 * nc) + 64;
        if (hipMalloc(&img, tot * 2) != hipSuccess || hipMalloc(&rc, (size_t)(nr + 4) * 4) != hipSuccess ||
            hipMalloc(&rs, (size_t)(nr + 4) * 4) != hipSuccess || hipMalloc(&ob, (size_t)B * N * D * 2 * 2 + 64) != hipSuccess)
            return E.fail("test_attention_mfma: alloc failed");
        (void)hipMemsetAsync(img, 0, tot * 2, ST(stream));
        (void)launch_rope_cossin(rope, rc, rs, nr, ST(stream));
        bf16_t* p = img;
        auto take = [&](size_t n) { bf16_t* r = p; p += (n + 7) & ~size_t(7); return r; };
        QkvPackArgs pk{};
        pk.qkvg = qkvg; pk.qw = qw; pk.kw = kw; pk.eps = eps; pk.q_scale = 1.0f / sqrtf((float)dh);
        pk.rope_cos = rc; pk.rope_sin = rs; pk.rot_dim = rot_dim; pk.prec = pa;
        pk.q = take(nqk); pk.q_lo = take(nqk); pk.k = take(nqk); pk.k_lo = take(nqk); pk.vt = take(nvt); pk.vt_lo = take(nvt);
        pk.g = take(ng); pk.g_lo = take(ng);
        pk.B = B; pk.N = N; pk.H = H; pk.dh = dh; pk.dhp = dhp; pk.Np = Np;
        hipError_t e = launch_qkv_pack(pk, ST(stream));
        AttnImg ai{};
        ai.prec = pa;
        ai.q = pk.q; ai.q_lo = pk.q_lo; ai.k = pk.k; ai.k_lo = pk.k_lo; ai.vt = pk.vt; ai.vt_lo = pk.vt_lo; ai.g = pk.g; ai.g_lo = sm_lo_for(pa, pk.g_lo);
        if (Cp > 0 && e == hipSuccess) {
            CrossPackArgs cp{};
            cp.k_ref = k_ref; cp.v_ref = v_ref; cp.k_text = k_text; cp.v_text = v_text;
            cp.kc = take(nc); cp.kc_lo = take(nc); cp.vtc = take(nc); cp.vtc_lo = take(nc);
            cp.prec = pa; cp.L = 1; cp.B = B; cp.H = H; cp.dh = dh; cp.dhp = dhp; cp.R = R > 0 ? R : 0; cp.P = P > 0 ? P : 0; cp.Rp = Rp; cp.Cp = Cp;
            e = launch_cross_pack(cp, ST(stream));
            ai.kc = cp.kc; ai.kc_lo = cp.kc_lo; ai.vtc = cp.vtc; ai.vtc_lo = cp.vtc_lo;
        }
        ai.mask_self = mask_self; ai.mask_ref = mask_ref; ai.mask_text = mask_text;
        ai.out_hi = ob; ai.out_lo = ob + (((size_t)B * N * D + 7) & ~size_t(7)); ai.ors = D;   // split pair: hi + lo ~ fp32
        ai.B = B; ai.N = N; ai.H = H; ai.dh = dh; ai.Np = Np; ai.R = R > 0 ? R : 0; ai.P = P > 0 ? P : 0; ai.Rp = Rp; ai.Cp = Cp;
        if (e == hipSuccess) e = launch_attention_img(ai, ST(stream));
        if (e == hipSuccess) e = launch_split_to_f32(ai.out_hi, ai.out_lo, out, (long)B * N * D, ST(stream));
        (void)hipStreamSynchronize(ST(stream));
        (void)hipFree(img); (void)hipFree(rc); (void)hipFree(rs); (void)hipFree(ob);
        return e == hipSuccess ? 0 : E.fail_hip(e, "attention_img");
    }
}

}  // extern "C"
