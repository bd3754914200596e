// Microbenchmark: what limits one wave's LDS-DMA issue rate on MI355X?
//   mode 0: M0 rewritten before every global_load_lds_dwordx4 (the gemm3/gemm4 pattern)
//   mode 1: M0 written once per 8 pieces, pieces addressed with the instruction's offset: field (-4096..3072)
//   mode 2: M0 never changes (all pieces to one slot; timing only)
//   mode 3: M0 saved to an SGPR, set, DMA, restored — the pattern gemm3 shipped with
// W waves per CU stream an L2-resident buffer; prints GB/s per wave and per CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, size_t bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)LPTR(smem) + wave * 16384;   // 16 KiB ring per wave
    size_t pos = ((size_t)blockIdx.x * nw + wave) * 8192 + lane * 16;
    const size_t stride = (size_t)gridDim.x * nw * 8192;
    for (int it = 0; it < iters; ++it) {
        const char* p = buf + (pos % bytes);
        pos += stride;
        const unsigned base = lds0 + (it & 1) * 8192;
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p + u * 1024), "s"(base + u * 1024) : "memory", "m0");
        } else if (MODE == 1) {
            const char* q = p + 4096;  // offsets -4096 .. 3072 around the middle
            asm volatile(
                "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %0, off offset:-4096\n\t"
                "global_load_lds_dwordx4 %0, off offset:-3072\n\t"
                "global_load_lds_dwordx4 %0, off offset:-2048\n\t"
                "global_load_lds_dwordx4 %0, off offset:-1024\n\t"
                "global_load_lds_dwordx4 %0, off\n\t"
                "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                "global_load_lds_dwordx4 %0, off offset:3072" ::"v"(q), "s"(base + 4096) : "memory", "m0");
        } else if (MODE == 3) {  // gemm3's original pattern: save M0, set, DMA, restore
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p + u * 1024), "s"(base + u * 1024) : "memory");
            }
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0) : "memory", "m0");
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(p + u * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // previous group landed, this one in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[lane] == 0x7f && lane == 63) *sink = 1;
}

int main() {
    const size_t bytes = 8u << 20;
    char* buf; hipMalloc(&buf, bytes + (1 << 20)); hipMemset(buf, 1, bytes + (1 << 20));
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int waves : {1, 2, 4, 8}) {
            const int iters = 4000;
            auto run = [&]() {
                const size_t lds = (size_t)waves * 16384;
                if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), lds, 0, buf, bytes, iters, sink); }
                if (mode == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), lds, 0, buf, bytes, iters, sink); }
                if (mode == 3) { hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<3>, dim3(256), dim3(waves * 64), lds, 0, buf, bytes, iters, sink); }
                if (mode == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<2>, dim3(256), dim3(waves * 64), lds, 0, buf, bytes, iters, sink); }
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_cu = (double)waves * iters * 8192.0 / (ms * 1e-3) / 1e9;
            printf("mode %d (%s)  waves/CU %d : %7.1f GB/s per CU  %6.1f GB/s per wave  %6.0f ns per 1-KiB piece per wave\n", mode,
                   mode == 0 ? "M0 per piece " : mode == 1 ? "M0 per 8, imm" : mode == 2 ? "M0 fixed     " : "M0 save/rest ", waves, per_cu, per_cu / waves,
                   ms * 1e6 / (iters * 8.0));
        }
    return 0;
}
