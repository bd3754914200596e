// Common device/host helpers for the smalltts gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define SMTTS_WAVE 64

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

// fp32 -> (bf16 hi, bf16 lo) with x ~= hi + lo
__device__ __forceinline__ void split1(float v, bf16_t& h, bf16_t& l) {
    h = (bf16_t)v;
    l = (bf16_t)(v - (float)h);
}
__device__ __forceinline__ void store_split4(bf16_t* hi, bf16_t* lo, long off, const float4& v) {
    bf16x4 h, l;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    l[0] = (bf16_t)(v.x - (float)h[0]); l[1] = (bf16_t)(v.y - (float)h[1]);
    l[2] = (bf16_t)(v.z - (float)h[2]); l[3] = (bf16_t)(v.w - (float)h[3]);
    *reinterpret_cast<bf16x4*>(hi + off) = h;
    if (lo) *reinterpret_cast<bf16x4*>(lo + off) = l;
}

// LDS pointer for direct-to-LDS DMA operands and a counted wait on this wave's outstanding vector-memory operations
#define SM_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
