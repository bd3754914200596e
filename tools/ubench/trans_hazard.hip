// Microbenchmark (r03, root cause of the fused q / k prep wrong-result mode, NOTEBOOK §13): does a VALU instruction that consumes the
// result of a transcendental instruction (v_rsq_f32) a few issue slots later read a STALE register in the last quarter of the
// wave (lanes 48..63) when ANOTHER wave on the same SIMD keeps the transcendental pipe busy?
//
// victim kernel : per iteration  r = v_rsq_f32(x);  <GAP independent VALU instructions>;  y = v_pk_mul_f32(v, r) (or v_mul_f32)
//                 and, as the reference, the same product taken after 16+ wait states.  Bitwise mismatches are counted per
//                 quarter wave (lanes 0-15 / 16-31 / 32-47 / 48-63).
// aggressor     : a long-running kernel on a second stream whose waves share the victim's SIMDs and issue only
//                 mode 0: nothing (no aggressor)   1: v_exp_f32 (transcendental)   2: v_fma_f32   3: v_mfma   4: v_pk_fma_f32
//
// hipcc --offload-arch=gfx950 -O3 tools/ubench/trans_hazard.hip -o /tmp/trans_hazard && /tmp/trans_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int GAP, bool PACKED>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    float x = in[threadIdx.x] + 1.0f;              // > 0
    f32x2 v = {in[threadIdx.x + 256], in[threadIdx.x + 512]};
    unsigned nbad = 0;
    float filler = x;
    for (int it = 0; it < iters; ++it) {
        x = x * 1.0009765625f + 0.03125f;          // a new argument every iteration
        float r, rs, yx, yy;
        // tight: consumer GAP VALU instructions after the transcendental (the compiler's own hazard handling adds what it
        // believes gfx950 needs: one wait state)
        asm volatile("v_rsq_f32 %0, %1" : "=v"(r) : "v"(x));
        if (GAP >= 1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(filler));
        if (GAP >= 2) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(filler));
        if (GAP >= 3) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(filler));
        if (GAP >= 4) asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0" : "+v"(filler));
        if (PACKED) {
            f32x2 rr = {r, r}, y2;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(v), "v"(rr));   // (the pair is built by the compiler: v_mov or op_sel)
            yx = y2.x; yy = y2.y;
        } else {
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %3" : "=&v"(yx), "=&v"(yy) : "v"(v.x), "v"(r), "v"(v.y));
        }
        // safe: the same values with a long gap
        asm volatile("v_rsq_f32 %0, %1\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=v"(rs) : "v"(x));
        float sx = v.x * rs, sy = v.y * rs;
        asm volatile("s_nop 7" : "+v"(sx), "+v"(sy));
        unsigned ax = __builtin_bit_cast(unsigned, yx), ay = __builtin_bit_cast(unsigned, yy);
        unsigned bx = __builtin_bit_cast(unsigned, sx), by = __builtin_bit_cast(unsigned, sy);
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(bx), "+v"(by));
        nbad += (ax != bx) ? 1u : 0u;
        nbad += (ay != by) ? 1u : 0u;
    }
    if (filler == 12345.f) nbad += 1;
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);
}

// The production pattern as the compiler emits it: rsq -> one VALU + two SALU -> v_pk_mul with op_sel broadcast
template <int DUMMY>
__global__ __launch_bounds__(256) void victim_compiled(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    float x = in[threadIdx.x] + 1.0f;
    f32x2 v = {in[threadIdx.x + 256], in[threadIdx.x + 512]};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1.0009765625f + 0.03125f;
        const float r = __builtin_amdgcn_rsqf(x);
        f32x2 y = v * r;
        asm volatile("" : "+v"(y));
        float rs = __builtin_amdgcn_rsqf(x);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(rs));
        f32x2 ys = {v.x * rs, v.y * rs};
        asm volatile("s_nop 7" : "+v"(ys));
        unsigned ax = __builtin_bit_cast(unsigned, y.x), ay = __builtin_bit_cast(unsigned, y.y);
        unsigned bx = __builtin_bit_cast(unsigned, ys.x), by = __builtin_bit_cast(unsigned, ys.y);
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(bx), "+v"(by));
        nbad += (ax != bx) ? 1u : 0u;
        nbad += (ay != by) ? 1u : 0u;
    }
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);
}

template <int MODE>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 0.5f, c = 0.25f, d = 0.125f;
    f32x2 p = {a, b}, q = {c, d};
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)a; fb[i] = (__bf16)b; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 2) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        } else if (MODE == 4) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %1, %1, %0, %0" : "+v"(p), "+v"(q));
        }
    }
    if (a + b + c + d + p.x + q.y + acc[0] == 12345.678f) out[threadIdx.x] = a;
}

template <class K>
static void run(const char* name, K kern, int mode, const float* in, unsigned* bad, hipStream_t sv, hipStream_t sa, float* sink) {
    CK(hipMemsetAsync(bad, 0, 16, sv));
    CK(hipStreamSynchronize(sv));
    const int agg_iters = 400000;   // a few ms
    // aggressor grid: 4 workgroups per CU next to one victim workgroup per CU -> victim waves share their SIMDs with 4 aggressor waves
    for (int rep = 0; rep < 4; ++rep) {
        switch (mode) {
            case 1: hipLaunchKernelGGL(aggressor<1>, dim3(1024), dim3(256), 0, sa, sink, agg_iters); break;
            case 2: hipLaunchKernelGGL(aggressor<2>, dim3(1024), dim3(256), 0, sa, sink, agg_iters); break;
            case 3: hipLaunchKernelGGL(aggressor<3>, dim3(1024), dim3(256), 0, sa, sink, agg_iters / 8); break;
            case 4: hipLaunchKernelGGL(aggressor<4>, dim3(1024), dim3(256), 0, sa, sink, agg_iters); break;
            default: break;
        }
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, sv, in, bad, 20000);
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
    }
    unsigned h[4];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    static const char* mn[] = {"no aggressor", "v_exp_f32 aggressor", "v_fma_f32 aggressor", "v_mfma aggressor", "v_pk_fma_f32 aggressor"};
    printf("%-34s | %-24s | mismatching products in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u\n", name, mn[mode], h[0], h[1], h[2], h[3]);
}

int main() {
    float* in; unsigned* bad; float* sink;
    CK(hipMalloc(&in, 768 * 4)); CK(hipMalloc(&bad, 16)); CK(hipMalloc(&sink, 1024));
    float h[768];
    for (int i = 0; i < 768; ++i) h[i] = 0.37f + 0.001f * i;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    for (int mode = 0; mode <= 4; ++mode) {
        run("compiled: rsq -> v_pk_mul", victim_compiled<0>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 0, v_pk_mul_f32", victim<0, true>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 1, v_pk_mul_f32", victim<1, true>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 2, v_pk_mul_f32", victim<2, true>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 3, v_pk_mul_f32", victim<3, true>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 5, v_pk_mul_f32", victim<4, true>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 0, v_mul_f32", victim<0, false>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 1, v_mul_f32", victim<1, false>, mode, in, bad, sv, sa, sink);
        run("asm: rsq, gap 2, v_mul_f32", victim<2, false>, mode, in, bad, sv, sa, sink);
    }
    return 0;
}
