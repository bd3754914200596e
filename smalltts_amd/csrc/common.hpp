// Common device/host helpers for the smalltts gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define SMTTS_WAVE 64
// Environment switches.  A shipped library reads TEN (SMTTS_ATTN_EPI, _GEMM_DEEP, _GEMM_GROUP, _GEMM_XCD, _LN_FOLD,
// _MIXER_STREAM, _MIXER_WIDE, _PERSIST_CUS, _SINGLE_STREAM, _STAGE_CHAIN: each selects between two product paths that
// tests/test_codec_gpu.py / test_kernels_gpu.py hold to each other).  Every other switch belongs to the A/B sessions under tools/ and is
// seen only by a lab build (make LAB=1 -> -DSMTTS_LAB): a stray variable cannot put the product into an unmeasured configuration.
#include <cstdlib>
static inline const char* lab_env(const char* name) {
#ifdef SMTTS_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// One-time launch setup PER DEVICE: hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count belong to a device, and a
// process may hold engines on several (smtts_create(device_id)).  Thread-safe: the bit is published after the setup
// succeeded; two threads racing on a new device at worst both run the idempotent setup.
struct DevOnce {
    std::atomic<unsigned> mask{0};
    int cus[16] = {};
    template <class F>
    hipError_t ensure(F&& setup, int* cu_out = nullptr) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 15;
        if (!((mask.load(std::memory_order_acquire) >> dev) & 1u)) {
            hipError_t e = setup();
            if (e != hipSuccess) return e;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            cus[dev] = n;
            mask.fetch_or(1u << dev, std::memory_order_release);
        }
        if (cu_out) {
            extern thread_local int g_persist_cus;   // engine.hip: cap on the grid of the persistent codec kernels (0 = every CU), installed per codec call
            *cu_out = g_persist_cus > 0 && g_persist_cus < cus[dev] ? g_persist_cus : cus[dev];
        }
        return hipSuccess;
    }
    int real_cus() const { int dev = 0; (void)hipGetDevice(&dev); return cus[dev & 15]; }   // (after ensure) the device's CU count, uncapped
};

// Row addressing shared by GEMM operands/outputs: logical row m -> element offset.
// rpb == 0: off + m*ld.  rpb > 0 (batched rows with per-batch padding):
// off + (m / rpb) * bstride + (m % rpb) * ld.
struct RowMap {
    long off;
    long ld;
    long bstride;
    int rpb;
    __host__ __device__ inline long at(int m) const {
        return rpb ? off + (long)(m / rpb) * bstride + (long)(m % rpb) * ld : off + (long)m * ld;
    }
    __host__ __device__ inline int batch(int m) const { return rpb ? m / rpb : 0; }
};
static inline RowMap rowmap_plain(long ld, long off = 0) { return RowMap{off, ld, 0, 0}; }
static inline RowMap rowmap_batched(long ld, int rpb, long bstride, long off) {
    return RowMap{off, ld, bstride, rpb};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {  // reduce within aligned groups of W lanes
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Cheap transcendentals for fused epilogues (the exact libm erff/expf/tanhf cost 50-150 VALU ops per
// element and made the codec FFN1 epilogue VALU-bound).  v_exp_f32 / v_rcp_f32 are ~1 ulp.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + __expf(-x)); }
// exact-erf GELU via Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): erfc(z) = t (a1 + t (a2 + ...)) exp(-z^2),
// t = 1 / (1 + p z), z = |x| / sqrt 2, and gelu(x) = x/2 + |x|/2 erf(z) — no compare / select, |.| are source modifiers.
__device__ __forceinline__ float gelu_f(float x) {
    const float t = fast_rcp(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));  // exp(-z^2)
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), 1.0f - poly * e, hx);
}
// GELU for values that are rounded to a 16-bit GEMM operand right afterwards (fp16: 4.9e-4 relative).  Two cheaper forms:
// * Gelu3: the three-term A&S 7.1.25 erfc (|gelu error| <= 1.25e-5 |x|): two fma fewer than gelu_f, still rcp + exp2;
// * GeluQ5: gelu(x) = max(x, 0) - |x| g(|x|) with g(a) = Phi(-a) = erfc(a / sqrt 2) / 2 = 2^q(a), q a degree-5 polynomial fitted
//   for the absolute error of a g(a) (tests/studies/gelu_q5_fit.py): |gelu error| <= 2.1e-6 for ALL x (fp32 evaluation, the
//   leading coefficient is negative so large |x| extrapolates to exactly max(x, 0)); 5 fma + exp2 + max + fma = 8
//   instructions with ONE transcendental and no reciprocal — against 13 (two transcendentals) for gelu_f.
struct Gelu3 {   // constants shared by the fused FFN kernels' hand-scheduled copies of this formula
    static constexpr float P = 0.47047f * 0.70710678118654752f, A1 = 0.3480242f, A2 = -0.0958798f, A3 = 0.7478556f;
};
__device__ __forceinline__ float gelu3_f(float x) {
    const float t = fast_rcp(fmaf(fabsf(x), Gelu3::P, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, Gelu3::A3, Gelu3::A2), Gelu3::A1);
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), 1.0f - poly * e, hx);
}
struct GeluQ5 {
    static constexpr float Q0 = -1.000138521194458f, Q1 = -1.1501970291137695f, Q2 = -0.46113213896751404f,
                           Q3 = -0.050880067050457f, Q4 = 0.006735440343618393f, Q5 = -0.00042676751036196947f;
};
// max(x, 0) in ONE instruction: fmaxf() makes hipcc canonicalise an operand it cannot prove quiet (an MFMA accumulator) with an
// extra v_max x, x
__device__ __forceinline__ float relu_f(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float gelu_q5_f(float x) {
    const float a = fabsf(x);
    float q = fmaf(GeluQ5::Q5, a, GeluQ5::Q4);
    q = fmaf(q, a, GeluQ5::Q3);
    q = fmaf(q, a, GeluQ5::Q2);
    q = fmaf(q, a, GeluQ5::Q1);
    q = fmaf(q, a, GeluQ5::Q0);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), relu_f(x));
}
// GeluQ5 on a PAIR of values in packed fp16 (round 4; tests/studies/gelu_f16_packed.py, tools/ubench/mfma_valu.hip KIND 6):
// the fused codec FFN kernels round the hidden to fp16 anyway, v_pk_fma_f16 handles two values per lane-instruction at the
// plain-VALU rate AND in the shadow of MFMAs (unlike v_pk_fma_f32), so the same formula costs 5 plain + 1 transcendental
// instruction per value instead of 9 + 1 (fp32 evaluation + convert + clamp).  Cost in audio: 67.8 -> 67.5 dB of decode SNR with
// it on every stage from C = 256 down (CPU emulation).  The input is NOT clamped: callers run only where the hidden's range
// is certified from the weights (Engine::certify_codec_ffn) — |h| < 65504 by construction.
//   front: h -> (hp = fp16 pair, ax = |hp|, e = 2^q(ax));   back: max(hp, 0) - ax * e, already the packed operand pair
__device__ __forceinline__ half2_t h2_splat(float c) { half2_t r = {(half_t)c, (half_t)c}; return r; }
// 2^q of both halves of a packed pair.  hipcc's lowering of exp2 on a half2 is v_exp_f16 + v_exp_f16_sdwa (high half, zero-padded) +
// v_pack_b32_f16; writing the high half in place (dst_unused:UNUSED_PRESERVE) needs no pack.  The s_nop is the wait state a
// transcendental's result needs before a VALU instruction reads it (the SDWA write preserves, i.e. reads, the low half) — the
// compiler's hazard recogniser does not look inside inline asm.  Consumers of the result must not be the very next instruction
// either (SDWA dst_sel forwarding): every caller has other work in between.
__device__ __forceinline__ unsigned exp2_pk_f16(unsigned q) {
    unsigned e;
    asm("v_exp_f16_e32 %0, %1\n\ts_nop 0\n\tv_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
        : "=&v"(e) : "v"(q));
    return e;
}
// (a, b) -> packed fp16 pair, round to nearest even, NO clamp: for values whose range is certified (Engine::certify_codec_ffn)
__device__ __forceinline__ unsigned cvt_pk_f16_raw(float a, float b) {
    f32x2_t v;
    v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
}
// The PACKED-fp16 evaluation (gelu_q5_pk_front / _back: the fused codec FFN kernels at the fp16 operand format) is limited by its own
// fp16 Horner steps more than by the fit.  Emulated in fp16 against exact GELU (tests/studies/gelu_f16_packed.py, tests/test_host_cpu.py):
//   degree | rms error, x ~ N(0, 1.5) | max relative error, |gelu| > 1e-3 | max relative error, 1e-3 < |x| < 0.25
//     5    |        3.10e-4          |             0.9e-2               |              1.3e-3
//     4    |        3.12e-4          |             1.0e-2               |              1.9e-3
//     3    |        3.19e-4          |             2.3e-2               |              7.2e-3   (2^Q0 = 0.4971: a +0.29 % |x| bias at small |x|)
//   (correctly rounded fp16 of the exact value: 2.12e-4 / 4.9e-4 / 4.9e-4.)
// Round 5 shipped degree 3 (two of eleven VALU instructions per value pair fewer; 2-4 % of the VALU-bound kernels, 0.6 % of a batch);
// its small-|x| bias is systematic, not noise.  Degree 4 has a POSITIVE leading coefficient: q(a) turns upward past
// a ~ 12 and 2^q overflows for hidden values of a few hundred — NaN audio on the outlier weights of tests/test_range_guard_gpu.py
// (certified bound 840) — so it would need a clamp of |x|, which costs the instruction it saves.  Degrees 5 and 3 end in a negative
// coefficient (large |x| extrapolates to exactly max(x, 0)).  Default since round 6: degree 5.  -DGELU_PK_DEG=3 / 4 remain as A/B builds.
#ifndef GELU_PK_DEG
#define GELU_PK_DEG 5
#endif
struct GeluQ4 {   // minimax fit of a Phi(-a) = a 2^q(a) over a in [0, 9] (tests/studies/gelu_q5_fit.py fit(4)): |error| <= 2.0e-5 in exact arithmetic
    static constexpr float Q0 = -1.0013247728347778f, Q1 = -1.1435197591781616f, Q2 = -0.47347283363342285f, Q3 = -0.04111006110906601f,
                           Q4 = 0.0033284714445471764f;
};
struct GeluQ3 {   // fit(3): |error| <= 1.8e-4 in exact arithmetic (at |x| ~ 0.145)
    static constexpr float Q0 = -1.008443832397461f, Q1 = -1.113277792930603f, Q2 = -0.5132908821105957f, Q3 = -0.02108863927423954f;
};
__device__ __forceinline__ void gelu_q5_pk_front(float a, float b, unsigned& hp, unsigned& axp, unsigned& ep) {
    f32x2_t v;
    v.x = a; v.y = b;
    const half2_t h = __builtin_convertvector(v, half2_t);
    hp = __builtin_bit_cast(unsigned, h);
    axp = hp & 0x7fff7fffu;
    const half2_t ax = __builtin_bit_cast(half2_t, axp);
#if GELU_PK_DEG == 5
    half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ5::Q5), ax, h2_splat(GeluQ5::Q4));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q3));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q2));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q1));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q0));
#elif GELU_PK_DEG == 4
    half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ4::Q4), ax, h2_splat(GeluQ4::Q3));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ4::Q2));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ4::Q1));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ4::Q0));
#else
    half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ3::Q3), ax, h2_splat(GeluQ3::Q2));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ3::Q1));
    q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ3::Q0));
#endif
    ep = exp2_pk_f16(__builtin_bit_cast(unsigned, q));
}
__device__ __forceinline__ unsigned gelu_q5_pk_back(unsigned hp, unsigned axp, unsigned ep) {
    const half2_t z = {(half_t)0.f, (half_t)0.f};
    const half2_t r = __builtin_elementwise_max(__builtin_bit_cast(half2_t, hp), z);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_fma(-__builtin_bit_cast(half2_t, axp), __builtin_bit_cast(half2_t, ep), r));
}
__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)); softplus with torch's threshold (20) for parity
    float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_MISH = 3 };
template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if (ACT == ACT_SILU) return silu_f(x);
    if (ACT == ACT_GELU) return gelu_f(x);
    if (ACT == ACT_MISH) return mish_f(x);
    return x;
}

// ---- GEMM operand precisions ------------------------------------------------------------------------------------------
// Every GEMM site runs in one of three operand formats (fp32 accumulation always); the engine picks one per site:
//   PREC_BF16X3  x = hi + lo, both bf16; acc += A_lo W_hi + A_hi W_lo + A_hi W_hi   (3 MFMAs, ~2^-17 operand error)
//   PREC_F16     one fp16 array per operand (11 significant bits, saturating convert), 1 MFMA
//   PREC_BF16    one bf16 array per operand, 1 MFMA
// Activation buffers are written by their producer in the format of the CONSUMING GEMM.  The format travels with the
// (hi, lo) pointer pair every producer already takes: lo == null -> bf16 single, lo == SM_F16_TAG -> fp16 single (16-bit
// storage is shared: an fp16 array is addressed through the same bf16_t* type), anything else -> split pair.
//   PREC_F16X2   (SITE_CODEC_CONV's ConvTranspose products only) A as one fp16 array, W as an fp16 hi + lo pair:
//                acc += A W_lo + A W_hi (2 MFMAs): the weights exact to ~22 bits, the activations rounded to 11
enum { PREC_BF16 = 1, PREC_F16 = 2, PREC_BF16X3 = 3, PREC_F16X2 = 4 };
// The fp16 tag: low two address bits == 2.  The REST of the tag, when non-zero, is the address of a saturation counter (unsigned,
// 4-byte aligned, device memory): every producer that clamps a value to +-65504 while writing this format adds to it (sat_note).
// Real lo arrays are 256-byte aligned workspace carve-outs (engine.hip Bump) or hipMalloc'ed packs, so their low bits are zero.
#define SM_F16_TAG (reinterpret_cast<bf16_t*>((uintptr_t)2))
__host__ __device__ __forceinline__ bool sm_is_f16(const bf16_t* lo) { return ((uintptr_t)lo & 3u) == 2u; }
__host__ __device__ __forceinline__ bool sm_is_split(const bf16_t* lo) { return lo != nullptr && !sm_is_f16(lo); }
__host__ __device__ __forceinline__ unsigned* sm_sat_counter(const bf16_t* lo) {
    return sm_is_f16(lo) ? reinterpret_cast<unsigned*>((uintptr_t)lo & ~(uintptr_t)3) : nullptr;
}
// lo pointer that tells a producer which format to write for a consumer of precision `prec` (fp16: + where to count clamps)
static inline bf16_t* sm_lo_for(int prec, bf16_t* lo, unsigned* sat_counter = nullptr) {
    // the tag lives in the low two address bits: a real lo array at an odd bf16 element offset (address = 2 mod 4) would be read as
    // "fp16, count clamps into the word next to it" by every producer.  All callers pass 256-byte aligned carve-outs; this makes
    // the invariant a check instead of a comment.
    if (prec == PREC_BF16X3 && ((uintptr_t)lo & 3u) != 0) {
        fprintf(stderr, "smalltts_hip: split-bf16 lo array %p is not 4-byte aligned (would alias the fp16 format tag)\n", (void*)lo);
        abort();
    }
    if (((uintptr_t)sat_counter & 3u) != 0) {
        fprintf(stderr, "smalltts_hip: saturation counter %p is not 4-byte aligned\n", (void*)sat_counter);
        abort();
    }
    return prec == PREC_BF16X3 ? lo
           : (prec == PREC_F16 || prec == PREC_F16X2) ? reinterpret_cast<bf16_t*>((uintptr_t)sat_counter | 2u)
                                                      : nullptr;
}

// fp16 range guard.  The conversions saturate instead of producing inf, which keeps a rare outlier from
// poisoning a softmax / residual row but is SILENT; `diff` collects raw ^ clamped of every converted pair, and sat_note() adds the
// number of lanes that clamped anything to the site's counter — one atomic per wave and call, on a path that is never taken with
// in-range data.  Call sat_note only where no load is consumed afterwards (stores and loads share vmcnt: NOTEBOOK §5a).
__device__ __forceinline__ unsigned cvt_pk_f16_sat(float a, float b, unsigned& diff) {
    f32x2_t v;
    v.x = a; v.y = b;
    const half2_t raw = __builtin_convertvector(v, half2_t);
    const half2_t mx = {(half_t)65504.f, (half_t)65504.f};
    const half2_t h = __builtin_elementwise_max(__builtin_elementwise_min(raw, mx), -mx);
    diff |= __builtin_bit_cast(unsigned, raw) ^ __builtin_bit_cast(unsigned, h);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void sat_note(unsigned diff, const bf16_t* tag) {
    if (__builtin_expect(diff != 0u, 0)) {
        unsigned* c = sm_sat_counter(tag);
        if (c) {
            const unsigned long long m = __ballot(1);   // the lanes inside this branch
            if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(c, (unsigned)__popcll(m));
        }
    }
}
// (a, b) -> packed fp16 pair, round to nearest even, saturating at +-65504 (v_cvt_pk_f16_f32 + v_pk_min/max_f16)
__device__ __forceinline__ unsigned cvt_pk_f16_sat(float a, float b) {
    unsigned d = 0;
    return cvt_pk_f16_sat(a, b, d);
}
// same for values bounded below (GELU / SiLU-gated outputs): only +inf can occur, one v_pk_min_f16
__device__ __forceinline__ unsigned cvt_pk_f16_satpos(float a, float b) {
    f32x2_t v;
    v.x = a; v.y = b;
    half2_t h = __builtin_convertvector(v, half2_t);
    const half2_t mx = {(half_t)65504.f, (half_t)65504.f};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(h, mx));
}
__device__ __forceinline__ unsigned short cvt_f16_sat(float a) {
    return (unsigned short)(cvt_pk_f16_sat(a, 0.f) & 0xffffu);
}

// fp32 -> (bf16 hi, bf16 lo) with x ~= hi + lo
__device__ __forceinline__ void split1(float v, bf16_t& h, bf16_t& l) {
    h = (bf16_t)v;
    l = (bf16_t)(v - (float)h);
}
// one activation value / four consecutive ones in the GEMM-operand format selected by `lo` (see above)
__device__ __forceinline__ void store_act1(bf16_t* hi, bf16_t* lo, long off, float v) {
    if (sm_is_f16(lo)) {
        unsigned d = 0;
        reinterpret_cast<unsigned short*>(hi)[off] = (unsigned short)(cvt_pk_f16_sat(v, 0.f, d) & 0xffffu);
        sat_note(d, lo);
        return;
    }
    bf16_t h, l;
    split1(v, h, l);
    hi[off] = h;
    if (lo) lo[off] = l;
}
__device__ __forceinline__ void store_split4(bf16_t* hi, bf16_t* lo, long off, const float4& v) {
    if (sm_is_f16(lo)) {
        uint2 p;
        unsigned d = 0;
        p.x = cvt_pk_f16_sat(v.x, v.y, d);
        p.y = cvt_pk_f16_sat(v.z, v.w, d);
        *reinterpret_cast<uint2*>(hi + off) = p;
        sat_note(d, lo);
        return;
    }
    bf16x4 h, l;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    l[0] = (bf16_t)(v.x - (float)h[0]); l[1] = (bf16_t)(v.y - (float)h[1]);
    l[2] = (bf16_t)(v.z - (float)h[2]); l[3] = (bf16_t)(v.w - (float)h[3]);
    *reinterpret_cast<bf16x4*>(hi + off) = h;
    if (lo) *reinterpret_cast<bf16x4*>(lo + off) = l;
}
// one 32x32x16 MFMA on 16-bit fragments held as bf16x8 registers: fp16 when SPLIT == PREC_F16, bf16 otherwise
template <int SPLIT>
__device__ __forceinline__ floatx16 mfma16(const bf16x8& a, const bf16x8& b, const floatx16& c) {
    if constexpr (SPLIT == PREC_F16 || SPLIT == PREC_F16X2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- attention operand images (attention_img.hip): element stores in format `prec` and the q / k head prep -------------------
__device__ __forceinline__ void store_img2(bf16_t* hi, bf16_t* lo, int prec, long off, float a, float b) {   // off even
    if (prec == PREC_F16) {   // (`lo` is not an array in this format: it carries the fp16 tag + saturation counter, or null)
        unsigned dd = 0;
        *reinterpret_cast<unsigned*>(hi + off) = cvt_pk_f16_sat(a, b, dd);
        sat_note(dd, lo);
        return;
    }
    bf16_t ah, al, bh, bl;
    split1(a, ah, al);
    split1(b, bh, bl);
    *reinterpret_cast<unsigned*>(hi + off) =
        (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    if (prec == PREC_BF16X3)
        *reinterpret_cast<unsigned*>(lo + off) =
            (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
}
__device__ __forceinline__ void store_img1(bf16_t* hi, bf16_t* lo, int prec, long off, float v) {
    if (prec == PREC_F16) {
        unsigned dd = 0;
        reinterpret_cast<unsigned short*>(hi)[off] = (unsigned short)(cvt_pk_f16_sat(v, 0.f, dd) & 0xffffu);
        sat_note(dd, lo);
    } else {
        bf16_t hh, ll;
        split1(v, hh, ll);
        hi[off] = hh;
        if (prec == PREC_BF16X3) lo[off] = ll;
    }
}
// the per-pair arithmetic shared by the epilogue and qkv_pack_kernel: (x0, x1) = dims (d, d + 1) of one head row, ss = the row's
// sum of squares.  Explicit fmaf: both call sites must round identically.
struct QkPrep {
    const float* w;        // [H][dh] norm weight of this part
    const float *rope_cos, *rope_sin;
    int rot_dim, dh;
    float eps, scale;      // scale: 1/sqrt(dh) for q, 1 for k
    // the loads of one pair, separated from the arithmetic so that a caller can issue them for all its rows first
    __device__ __forceinline__ void weights(int h, int d, float& w0, float& w1) const {
        const bool in = d < dh;
        w0 = in ? w[h * dh + d] : 0.f;
        w1 = in ? w[h * dh + d + 1] : 0.f;
    }
    __device__ __forceinline__ void rope(int d, int n, float& c, float& s) const {
        c = 1.f; s = 0.f;
        if (d < rot_dim) { c = rope_cos[(long)n * rot_dim + d]; s = rope_sin[(long)n * rot_dim + d]; }
    }
    __device__ __forceinline__ void math(float& x0, float& x1, float ss, float w0, float w1, float c, float s, int d) const {
        const float rstd = __builtin_amdgcn_rsqf(fmaf(ss, 1.0f / (float)dh, eps));
        x0 = (x0 * rstd) * w0;
        x1 = (x1 * rstd) * w1;
        if (d < rot_dim) {
            const float a0 = fmaf(x0, c, -(x1 * s)), a1 = fmaf(x1, c, x0 * s);
            x0 = a0; x1 = a1;
        }
        x0 *= scale; x1 *= scale;
    }
    __device__ __forceinline__ void apply(float& x0, float& x1, float ss, int h, int d, int n) const {
        float w0, w1, c, s;
        weights(h, d, w0, w1);
        rope(d, n, c, s);
        math(x0, x1, ss, w0, w1, c, s, d);
    }
};

// LDS pointer for direct-to-LDS DMA operands and a counted wait on this wave's outstanding vector-memory operations
#define SM_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
