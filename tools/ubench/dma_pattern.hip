// Microbenchmark: LDS-DMA piece cost vs the global access pattern of a GEMM tile (MI355X).
// A piece = one global_load_lds_dwordx4 (64 lanes x 16 B).  Patterns:
//   0 contiguous 1 KiB
//   1 8 rows x 128 B, row stride `ld` bytes, lanes in row order
//   2 as 1 with the 16-B chunks of each row XOR-permuted by (row>>1)&7   (gemm3/gemm4 swizzle)
//   3 16 rows x 64 B, row stride ld, chunk ^ ((row>>2)&3)                 (BK=32 variant)
//   4 as 1 but the two 64-B halves of a row swapped by row parity (quad-granular permutation)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_pattern.hip -o tools/ubench/bin/dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, size_t bytes, int iters, int pattern, int ld, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)LPTR(smem) + wave * 16384;
    long lane_off;
    if (pattern == 0) lane_off = lane * 16;
    else if (pattern == 1) lane_off = (long)(lane >> 3) * ld + (lane & 7) * 16;
    else if (pattern == 2) { const int r = lane >> 3; lane_off = (long)r * ld + (((lane & 7) ^ ((r >> 1) & 7))) * 16; }
    else if (pattern == 3) { const int r = lane >> 2; lane_off = (long)r * ld + (((lane & 3) ^ ((r >> 2) & 3))) * 16; }
    else { const int r = lane >> 3; lane_off = (long)r * ld + (((lane & 7) ^ ((r & 1) << 2))) * 16; }
    const long piece_step = pattern == 0 ? 1024 : (pattern == 3 ? 16L * ld : 8L * ld);  // next 8 (16) rows
    // each wave walks its own row panel: 8 pieces down the rows, then the next 128-B k-slice
    size_t wbase = ((size_t)blockIdx.x * nw + wave) * 8 * (size_t)piece_step;
    size_t kofs = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned base = lds0 + (it & 1) * 8192;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* p = buf + ((wbase + u * piece_step + kofs + lane_off) & (bytes - 1));  // bytes is a power of two
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(base + u * 1024) : "memory", "m0");
        }
        kofs += pattern == 0 ? (size_t)gridDim.x * nw * 8192 : (pattern == 3 ? 64 : 128);
        if (pattern != 0 && kofs >= (size_t)ld) { kofs = 0; wbase += (size_t)gridDim.x * nw * 8 * piece_step; }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[lane] == 0x7f && lane == 63) *sink = 1;
}

int main() {
    const size_t bytes = 2u << 20;  // L2-resident (4 MiB L2 per XCD); power of two
    char* buf; hipMalloc(&buf, bytes + (32 << 20)); hipMemset(buf, 1, bytes + (32 << 20));
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"contiguous 1 KiB      ", "8 rows x128B in order ", "8 rows x128B xor-swz  ", "16 rows x64B xor-swz  ", "8 rows x128B half-swap"};
    for (int ld : {1920, 8192})
        for (int pattern = 0; pattern < 5; ++pattern)
            for (int waves : {4, 8}) {
                const int iters = 2000;
                const size_t lds = (size_t)waves * 16384;
                hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                auto run = [&]() { hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), lds, 0, buf, bytes, iters, pattern, ld, sink); };
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double per_cu = (double)waves * iters * 8192.0 / (ms * 1e-3) / 1e9;
                printf("ld %5d  %s waves/CU %d : %7.1f GB/s per CU  %6.0f ns per piece per wave\n", ld, names[pattern], waves, per_cu,
                       ms * 1e6 / (iters * 8.0));
            }
    return 0;
}
