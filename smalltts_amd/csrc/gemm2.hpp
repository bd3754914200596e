// GEMM v2 for gfx950: multi-stage LDS ring filled by direct-to-LDS DMA (global_load_lds_dwordx4),
// XOR-swizzled through the per-lane SOURCE address, counted vmcnt + raw s_barrier (one barrier per
// k-tile), fp32 A split to bf16 hi/lo at fragment-read time.  Same math / epilogue functors as
// gemm.hpp; requires K % 64 == 0 (the engine pads the one odd K, FF2's 2400 -> 2432).
//
// Why: v1 keeps one k-tile in flight per workgroup and the DiT GEMMs (M = 600) run ~1 workgroup
// per CU, so every k-step paid a full HBM round trip (~1.7 us).  Here S-1 tiles (32..48 KB each) are
// in flight per CU with no staging registers and no ds_write pass.
//
// LDS image of one stage (lane-linear per DMA instruction = 1 KiB):
//   A   : BM rows x 256 B (64 fp32); 16-B chunk c of row r lives at chunk position c ^ (r & 15)
//   Whi : BN rows x 128 B (64 bf16); chunk c of row r at position c ^ ((r >> 1) & 7)      (Wlo same)
// -> every ds_read_b128 fragment read (16 lanes = 16 distinct rows mod 16) is bank-conflict free.
#pragma once
#include "gemm.hpp"

#define SM_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SM_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int SPLIT, int S, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(GemmOperands g, Epi epi) {
    constexpr int BK = 64;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_BYTES = BM * 256, W_BYTES = BN * 128;
    constexpr int NWARR = SPLIT == 3 ? 2 : 1;
    constexpr int STAGE = A_BYTES + NWARR * W_BYTES;
    constexpr int A_PW = (BM / 4) / NW;  // DMA instructions per wave per stage for A
    constexpr int W_PW = (BN / 8) / NW;  // ... for each W array
    constexpr int DPS = A_PW + NWARR * W_PW;
    static_assert(A_PW * NW * 4 == BM && W_PW * NW * 8 == BN, "tile / wave count mismatch");
    static_assert(!Epi::PAIRED || TN == 2, "paired epilogue needs a 32x64 wave tile");
    static_assert((S - 2) * DPS <= 63, "vmcnt immediate overflow");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;

    // ---- per-lane DMA source pointers (k0 = 0) ------------------------------------------------
    const float* asrc[A_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int r = (wave * A_PW + i) * 4 + (lane >> 4), p = lane & 15;
        const int c = p ^ (r & 15);
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        asrc[i] = g.A + (long)z * g.a_z + g.amap.at(m) + c * 4;
    }
    const long wz = (long)(g.w_zmod ? z % g.w_zmod : z) * g.w_z;
    const bf16_t* wsrc[W_PW];
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int r = (wave * W_PW + i) * 8 + (lane >> 3), p = lane & 7;
        const int c = p ^ ((r >> 1) & 7);
        int n = n0 + r;
        n = n < g.N ? n : g.N - 1;
        wsrc[i] = g.Whi + wz + (long)n * g.ldw + c * 8;
    }
    const long lo_delta = SPLIT == 3 ? (g.Wlo - g.Whi) : 0;

    // DMA issue through inline asm: hipcc treats a builtin LDS-DMA as a pending LDS store and puts
    // `s_waitcnt vmcnt(0)` in front of the next ds_read (draining the ring every k-step); an asm DMA is
    // invisible to that bookkeeping, so completion is counted by hand (wait_vmcnt below).  M0 (LDS
    // destination base) is saved/restored inside the statement (cdna guide 5.7).
    auto dma16 = [&](const void* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    };
    const unsigned lds0 = (unsigned)(size_t)SM_LPTR(smem);
    auto issue = [&](int kt) {
        const unsigned st = lds0 + (unsigned)((kt % S) * STAGE);
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) dma16(asrc[i] + k0, st + (unsigned)((wave * A_PW + i) * 1024));
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            dma16(wsrc[i] + k0, st + (unsigned)(A_BYTES + (wave * W_PW + i) * 1024));
            if (SPLIT == 3) dma16(wsrc[i] + lo_delta + k0, st + (unsigned)(A_BYTES + W_BYTES + (wave * W_PW + i) * 1024));
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing (constant over k-tiles)
    const int fr = lane & 31, fh = lane >> 5;
    int a_row_off[TM], a_sw[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + fr;
        a_row_off[i] = r * 256;
        a_sw[i] = r & 15;
    }
    int w_row_off[TN], w_sw[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + fr;
        w_row_off[j] = r * 128;
        w_sw[j] = (r >> 1) & 7;
    }

    const int nk = g.K / BK;
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s);

    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most (S-2) younger stages of THIS wave are still in flight
        if (kt + S - 1 <= nk)
            wait_vmcnt<(S - 2) * DPS>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // everyone's DMA for stage kt landed; everyone finished reading stage kt-1
        if (kt + S - 1 < nk) issue(kt + S - 1);  // overwrites the buffer of stage kt-1
        const char* st = smem + (kt % S) * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int c0 = kk * 4 + fh * 2;
                const float4 f0 = *reinterpret_cast<const float4*>(st + a_row_off[i] + ((c0 ^ a_sw[i]) << 4));
                const float4 f1 = *reinterpret_cast<const float4*>(st + a_row_off[i] + (((c0 + 1) ^ a_sw[i]) << 4));
                const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ah[i][e] = (bf16_t)f[e];
                    if (SPLIT == 3) al[i][e] = (bf16_t)(f[e] - (float)ah[i][e]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = kk * 2 + fh;
                const int off = A_BYTES + w_row_off[j] + ((c ^ w_sw[j]) << 4);
                bh[j] = *reinterpret_cast<const bf16x8*>(st + off);
                if (SPLIT == 3) bl[j] = *reinterpret_cast<const bf16x8*>(st + off + W_BYTES);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    gemm_epilogue<TM, TN, Epi>(epi, acc, g.M, g.N, m0 + wm * TM * 32, n0 + wn * TN * 32, z, lane);
}

template <int BM, int BN, int WM, int WN, int SPLIT, int S, class Epi>
static inline hipError_t gemm2_launch_cfg(const GemmOperands& g, const Epi& epi, int Z, hipStream_t st) {
    constexpr int NWARR = SPLIT == 3 ? 2 : 1;
    constexpr size_t lds = (size_t)S * (BM * 256 + NWARR * BN * 128);
    static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, Z);
    auto kern = gemm2_kernel<BM, BN, WM, WN, SPLIT, S, Epi>;
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, g, epi);
    return hipGetLastError();
}

// v2 is usable when K is a whole number of 64-wide k-tiles and operands are 16-B aligned
static inline bool gemm2_ok(const GemmOperands& g) {
    return g.K % 64 == 0 && g.K >= 64 && (g.amap.ld % 4) == 0 && (g.amap.off % 4) == 0 && (g.ldw % 8) == 0;
}

template <int SPLIT, class Epi>
static inline hipError_t gemm2_launch_split(const GemmOperands& g, const Epi& epi, int Z, int cfg, hipStream_t st) {
    switch (cfg) {
        case CFG_64x128:
            return gemm2_launch_cfg<64, 128, 2, 2, SPLIT, 3, Epi>(g, epi, Z, st);
        case CFG_128x128:
            if constexpr (!Epi::PAIRED) return gemm2_launch_cfg<128, 128, 2, 2, SPLIT, 2, Epi>(g, epi, Z, st);
            break;
        case CFG_64x64:
            if constexpr (!Epi::PAIRED) return gemm2_launch_cfg<64, 64, 2, 2, SPLIT, 4, Epi>(g, epi, Z, st);
            break;
        case CFG_128x32:
            if constexpr (!Epi::PAIRED) return gemm2_launch_cfg<128, 32, 4, 1, SPLIT, 3, Epi>(g, epi, Z, st);
            break;
        case CFG_128x64:
            if constexpr (!Epi::PAIRED) return gemm2_launch_cfg<128, 64, 4, 1, SPLIT, 3, Epi>(g, epi, Z, st);
            break;
    }
    return hipErrorInvalidValue;
}

// Dispatcher used by the op wrappers: v2 when eligible, otherwise the v1 kernel.
extern int g_gemm_force_v1;  // test hook (0 = auto)
template <class Epi>
static inline hipError_t gemm_dispatch(const GemmOperands& g, const Epi& epi, int Z, int split, hipStream_t st,
                                       int cfg = -1) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (cfg < 0) cfg = gemm_pick_cfg(g.M, g.N, g.K, Epi::PAIRED);
    if (!g_gemm_force_v1 && gemm2_ok(g)) {
        if (split == 3) return gemm2_launch_split<3, Epi>(g, epi, Z, cfg, st);
        return gemm2_launch_split<1, Epi>(g, epi, Z, cfg, st);
    }
    return gemm_launch(g, epi, Z, split, st, cfg);
}
