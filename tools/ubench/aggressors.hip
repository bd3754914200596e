// Aggressor kernels for the r03 root-cause session (tools/stress_prep.py aggr): each keeps ONE kind of CU resource busy on a
// second stream while the fused-prep attention kernel runs next to it, to find out which neighbour activity makes its results
// stop repeating.  Built as a tiny shared library with extern "C" launchers (ctypes):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/aggressors.hip -o tools/ubench/bin/libaggr.so
// kinds: 0 LDS-DMA (global_load_lds_dwordx4 into the workgroup's own LDS)   1 ds_read_b128 + v_mfma   2 v_exp_f32
//        3 ds_write_b128 + s_barrier   4 global_load_dwordx4 into VGPRs   5 v_pk_fma_f32   6 DPP row ops   7 v_mfma only
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void aggr_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters, long n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a = tid * 1e-3f, b = 0.5f;
    f32x2 p = {a, b}, q = {b, a};
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)a; fb[i] = (__bf16)b; }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) void*)smem);
    long off = ((long)blockIdx.x * 256 + tid) * 4 % n;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            const float* g = src + off;
            off = (off + 256 * 4 * 1024) % n;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(wave * 1024 + (it & 7) * 4096)))) : "memory");
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (KIND == 1) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + it * 1024) & 32767));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, fb, acc, 0, 0, 0);
        } else if (KIND == 2) {
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(a), "+v"(b));
        } else if (KIND == 3) {
            *reinterpret_cast<float4*>(smem + ((tid * 16 + it * 4096) & 32767)) = make_float4(a, b, a, b);
            __builtin_amdgcn_s_barrier();
        } else if (KIND == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + off);
            off = (off + 256 * 4 * 1024) % n;
            a += v.x + v.w;
        } else if (KIND == 5) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %1, %1, %0, %0" : "+v"(p), "+v"(q));
        } else if (KIND == 6) {
            a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x140, 0xf, 0xf, false));
            b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x141, 0xf, 0xf, false));
        } else if (KIND == 7) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        } else if (KIND == 8) {
            auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
            a = __builtin_bit_cast(float, r[0]); b = __builtin_bit_cast(float, r[1]);
        } else if (KIND == 9) {   // the gemm3 k-loop shape: LDS-DMA ring + ds_read_b128 + mfma + raw barrier
            const float* g = src + off;
            off = (off + 256 * 4 * 1024) % n;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(wave * 1024 + (it & 7) * 4096)))) : "memory");
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + it * 1024) & 32767));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, fb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, x, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a + b + p.x + q.y + acc[0] + acc[7] == 12345.678f) sink[tid] = a;
}

extern "C" int aggr_launch(int kind, void* stream, const float* src, float* sink, int iters, long n, int blocks, int lds_bytes) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 g(blocks), t(256);
    switch (kind) {
        case 0: hipLaunchKernelGGL(aggr_kernel<0>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 1: hipLaunchKernelGGL(aggr_kernel<1>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 2: hipLaunchKernelGGL(aggr_kernel<2>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 3: hipLaunchKernelGGL(aggr_kernel<3>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 4: hipLaunchKernelGGL(aggr_kernel<4>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 5: hipLaunchKernelGGL(aggr_kernel<5>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 6: hipLaunchKernelGGL(aggr_kernel<6>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 7: hipLaunchKernelGGL(aggr_kernel<7>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 8: hipLaunchKernelGGL(aggr_kernel<8>, g, t, lds_bytes, st, src, sink, iters, n); break;
        case 9: hipLaunchKernelGGL(aggr_kernel<9>, g, t, lds_bytes, st, src, sink, iters, n); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
