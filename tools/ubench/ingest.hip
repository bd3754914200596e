// Microbenchmark: per-CU ingest bandwidth from an L2/MALL-resident buffer on MI355X.
//   mode 0: global_load_dwordx4 -> VGPR (8 loads in flight per lane)
//   mode 1: LDS-DMA global_load_lds_dwordx4 into a ring of `depth` KiB slots per wave (counted vmcnt)
// Build/run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/ingest.hip -o /tmp/ingest && /tmp/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int UNR>
__global__ __launch_bounds__(512) void k_vgpr(const uint4* __restrict__ buf, size_t n16, int iters, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { v[u] = buf[i % n16]; i += stride; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const uint4* __restrict__ buf, size_t n16, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)LPTR(smem) + wave * DEPTH * 1024;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const uint4* p = buf + (i % n16);
            i += stride;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p), "s"(lds0 + u * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");  // keep half the ring in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[lane] == 0x7f && lane == 63) *sink = 1;
}

int main() {
    const size_t sizes[] = {2u << 20, 16u << 20, 512u << 20};  // L2-resident, MALL-resident, HBM
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t bytes : sizes) {
        uint4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        const size_t n16 = bytes / 16;
        for (int mode = 0; mode < 3; ++mode) {
            for (int wgs : {256, 512}) {
                const int iters = 2000;
                auto run = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k_vgpr<8>, dim3(wgs), dim3(512), 0, 0, buf, n16, iters / 8, sink);
                    else if (mode == 1) { hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8 * 1024);
                        hipLaunchKernelGGL(k_dma<8>, dim3(wgs), dim3(512), 8 * 8 * 1024, 0, buf, n16, iters / 8, sink); }
                    else { hipFuncSetAttribute((const void*)k_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 1024);
                        hipLaunchKernelGGL(k_dma<16>, dim3(wgs), dim3(wgs == 512 ? 512 : 512), 8 * 16 * 1024, 0, buf, n16, iters / 16, sink); }
                };
                if (mode == 2 && wgs == 512) continue;  // 2 x 128 KiB does not fit one CU
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double moved = (double)wgs * 512 * 16.0 * iters;
                printf("buf %4zu MiB  mode %s  wgs %3d : %7.1f GB/s total  %6.1f GB/s per CU\n", bytes >> 20,
                       mode == 0 ? "vgpr x8     " : mode == 1 ? "lds-dma d8  " : "lds-dma d16 ", wgs, moved / ms / 1e6, moved / ms / 1e6 / 256);
            }
        }
        hipFree(buf);
    }
    return 0;
}
