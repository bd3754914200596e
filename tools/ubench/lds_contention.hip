// Microbenchmark: do LDS-DMA landings (global_load_lds_dwordx4) and ds_read_b128 streams share LDS bandwidth on MI355X?
// One 512-thread workgroup per CU.  Waves 0..R-1 run a conflict-free ds_read_b128 loop over a 64 KiB region, waves R..R+D-1
// stream an L2-resident buffer into another 64 KiB region by LDS-DMA.  Prints the rate of each stream alone and together.
// Extra modes: `mf` waves run dependent / independent v_mfma_f32_32x32x16_bf16 chains (mode 1: one accumulator, mode 2: four),
// optionally fed by the ds_reads (mode 3 = gemm-like: 6 reads + 6 MFMAs per step).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_contention.hip -o /tmp/lds_contention && /tmp/lds_contention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void k(const char* __restrict__ buf, size_t bytes, int iters, int readers, int dmas, unsigned* sink,
                                         unsigned long long* cycles, int mf, int mfmode, int prio, int bar) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = wall_clock64();
    if (wave < readers) {
        // 16 reads per loop trip, lane l reads 16 B at l*16 (+ rotating 1 KiB blocks): conflict free
        unsigned acc = 0;
        const char* base = smem + lane * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint4 v = *reinterpret_cast<const uint4*>(base + (((it * 16 + u) & 63) << 10));
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
        if (acc == 0x12345u) *sink = acc;
    } else if (wave < readers + dmas) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        const int d = wave - readers;
        const unsigned lds0 = (unsigned)(size_t)LPTR(smem) + 65536 + d * 8192;
        size_t pos = ((size_t)blockIdx.x * dmas + d) * 8192 + lane * 16;
        const int dit = bar ? iters : (mf ? iters * 6 : iters);
        for (int it = 0; it < dit; ++it) {
            if (bar) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const char* p2 = buf + ((pos + 4096) & (bytes - 1));   // second half of the 16 pieces
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p2 + u * 1024), "s"(lds0 + 32768 + u * 1024) : "memory", "m0");
            }
            const char* p = buf + (pos & (bytes - 1));
            pos += (size_t)gridDim.x * dmas * 8192;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p + u * 1024), "s"(lds0 + u * 1024) : "memory", "m0");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else if (wave < readers + dmas + mf) {
        floatx16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
        const char* base = smem + lane * 16;
        for (int it = 0; it < iters; ++it) {
            if (bar) __builtin_amdgcn_s_barrier();
            if (mfmode == 3) {  // gemm-like: 6 fragment reads + 6 MFMAs (two accumulators) per step, 4 steps
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bf16x8 f[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) f[q] = *reinterpret_cast<const bf16x8*>(base + (((it * 24 + u * 6 + q) & 63) << 10));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[2], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[3], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[2], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[4], acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[5], acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[4], acc[1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 24; ++u) {
                    const int j = mfmode == 2 ? (u & 3) : 0;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
        }
        float s = 0.f;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        if (s == 1.2345f) *sink = 1;
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

int main() {
    const size_t bytes = (size_t)(getenv("UB_MB") ? atoi(getenv("UB_MB")) : 2) << 20;  // power of two: 2 = L2-resident, 512 = HBM
    char* buf; hipMalloc(&buf, bytes + (1 << 20)); hipMemset(buf, 1, bytes + (1 << 20));
    unsigned* sink; hipMalloc(&sink, 4);
    unsigned long long* cyc; hipMalloc(&cyc, 128);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    struct Cfg { int r, d, mf, mode, prio; } cfgs[] = {{4, 0, 0, 0, 0}, {0, 4, 0, 0, 0}, {4, 4, 0, 0, 0}, {0, 0, 8, 3, 0}, {0, 4, 8, 3, 0},
                                                       {0, 4, 8, 3, 1}, {0, 2, 8, 3, 0}, {0, 4, 4, 3, 0}, {0, 0, 4, 3, 0},
                                                       {0, 0, 8, 3, 2}, {0, 4, 8, 3, 2}};  // prio 2 = one s_barrier per k-tile-sized step
    for (auto c : cfgs) {
        const int threads = (c.r + c.d + c.mf) * 64;
        auto run = [&]() { hipLaunchKernelGGL(k, dim3(256), dim3(threads), 128 * 1024 + 8192, 0, buf, bytes, iters, c.r, c.d, sink, cyc, c.mf, c.mode, c.prio == 1, c.prio == 2); };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double rd = (double)c.r * iters * 16 * 1024.0 / (ms * 1e-3) / 1e9;
        const double dm = (double)c.d * iters * 8192.0 / (ms * 1e-3) / 1e9;
        const double tf = (double)c.mf * iters * 24 * 32768.0 * 256 / (ms * 1e-3) / 1e12;  // chip TFLOP/s of the MFMA waves
        unsigned long long hc[12] = {0};
        hipMemcpy(hc, cyc, sizeof(unsigned long long) * (c.r + c.d + c.mf), hipMemcpyDeviceToHost);
        // per-stream rates from the waves' own elapsed time (block 0; s_memtime ticks at 100 MHz): streams finish at different times
        const double t_r = c.r ? hc[0] * 10e-9 : 0, t_d = c.d ? hc[c.r] * 10e-9 : 0, t_m = c.mf ? hc[c.r + c.d] * 10e-9 : 0;
        printf("readers %d dma %d mfma %d (mode %d, prio %d): %6.3f ms | ds_read %6.1f GB/s/CU | lds-dma %6.1f GB/s/CU | mfma %6.1f TF/s chip-equivalent\n",
               c.r, c.d, c.mf, c.mode, c.prio, ms, t_r ? c.r * iters * 16384.0 / t_r / 1e9 : 0.0, t_d ? c.d * ((c.mf && c.prio != 2) ? 6.0 : (c.prio == 2 ? 2.0 : 1.0)) * iters * 8192.0 / t_d / 1e9 : 0.0,
               t_m ? c.mf * iters * 24 * 32768.0 * 256 / t_m / 1e12 : 0.0);
    }
    return 0;
}
