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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)); softplus with torch's threshold (20) for parity
    float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_MISH = 3 };
template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if (ACT == ACT_SILU) return x / (1.0f + expf(-x));
    if (ACT == ACT_GELU) return gelu_f(x);
    if (ACT == ACT_MISH) return mish_f(x);
    return x;
}
