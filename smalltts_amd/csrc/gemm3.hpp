// GEMM v3 for gfx950 — the hot-path kernel.
//   C[M,N] = A[M,K] x W[N,K]^T with BOTH operands already split into bf16 hi/lo arrays in HBM:
//   weights at smtts_finalize, activations by the epilogue / norm kernel that produced them (each
//   element is split exactly once, instead of once per consuming workgroup as in v1/v2).
// * all four tile arrays (Ahi, Alo, Whi, Wlo; [rows][64] bf16 = 128 B rows) arrive by direct-to-LDS
//   DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into an S-stage ring; the XOR swizzle
//   (16-B chunk c of row r at position c ^ ((r>>1)&7)) is applied on the per-lane SOURCE address so
//   every ds_read_b128 fragment read is bank-conflict free;
// * the k-loop is ds_read_b128 + v_mfma_f32_32x32x16_bf16 only (no VALU conversion, no ds_write),
//   one raw s_barrier per k-tile, counted s_waitcnt vmcnt so S-1 tiles stay in flight;
// * 8 waves (2 per SIMD) per workgroup so one wave's LDS latency hides under the other's MFMAs.
// Requires K % 64 == 0, 16-B aligned rows.  Epilogue functors are shared with gemm.hpp.
#pragma once
#include "gemm.hpp"

struct Gemm3Operands {
    const bf16_t* Ahi;
    const bf16_t* Alo;
    RowMap amap;  // element offsets into Ahi / Alo
    const bf16_t* Whi;
    const bf16_t* Wlo;
    long ldw;
    int M, N, K;
    long a_z, w_z;
    int w_zmod;
    int nfast = 0;     // tile order inside an XCD's run: 0 = M fastest, 1 = N fastest, G > 1 = bands of G row tiles (set by gemm3_launch)
    int stage16 = 1;   // 16-bit outputs through the LDS-staged epilogue (set by gemm3_launch from g_gemm3_stage16; A/B switch)
    int ksplit_tiles;  // > 0: blockIdx.z is a split-K index; this launch slice covers k-tiles [z*ksplit_tiles, +ksplit_tiles)
};

template <int BM, int BN, int WM, int WN, int SPLIT, int S, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void gemm3_kernel(Gemm3Operands g, Epi epi) {
    constexpr int BK = 64;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    // arrays per operand: split-bf16 (SPLIT 3) has hi + lo on both sides; PREC_F16X2 (SPLIT 4) one fp16 array of A against an
    // fp16 hi + lo pair of W (two MFMAs per fragment pair: A W_lo + A W_hi); the single formats one each
    constexpr int NARR_A = SPLIT == 3 ? 2 : 1, NARR_W = (SPLIT == 3 || SPLIT == PREC_F16X2) ? 2 : 1;
    constexpr int NARR = NARR_A;
    constexpr int A_ARR = BM * 128, W_ARR = BN * 128;  // bytes per array per stage
    constexpr int STAGE = NARR_A * A_ARR + NARR_W * W_ARR;
    constexpr int NA = NARR_A * (BM / 8), NWS = NARR_W * (BN / 8);  // DMA slots (8 rows each)
    constexpr int PW = (NA + NWS + NW - 1) / NW;                // slots per wave per stage
    constexpr bool DUMMY = PW * NW != NA + NWS;                 // odd wave counts (160-row tiles, 10 waves): the surplus slots DMA
    constexpr int STAGE_LD = STAGE + (DUMMY ? 1024 : 0);        // into a 1-KiB pad behind the stage so every wave counts the same vmcnt
    static_assert(!Epi::PAIRED || TN == 2, "paired epilogue needs a 32x64 wave tile");
    // L2 prefetch-touch: one dword load per 128-B line of the tile PFD k-tiles beyond the stage being DMA'd.  HBM
    // has spare bandwidth (the loop is bound by bytes-in-flight / latency), so touching early makes the later DMA an
    // L2 hit.  Issued from every wave right after each stage's DMA so the vmcnt arithmetic stays uniform.
    // Measured on MI355X: no gain (dit.qkvg 36.5 -> 42 us), i.e. HBM-miss latency is not what binds -> disabled (PFD = 0).
    constexpr int PFD = 0;
    constexpr int LINES = NARR_A * BM + NARR_W * BN;
    constexpr int PFN = PFD ? (LINES + NW * 64 - 1) / (NW * 64) : 0;  // touch instructions per wave per k-tile
    constexpr int OPS = PW + PFN;                            // VMEM ops per wave per k-tile
    static_assert((S - 2) * PW + (S - 1) * PFN <= 63, "vmcnt immediate overflow");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (workgroup p runs on XCD p % 8, each XCD has its own L2): give every XCD a
    // contiguous run of virtual tile ids so tiles that share an operand panel hit the same L2 instead of
    // re-fetching it on 8 XCDs.
    const int Mt = (g.M + BM - 1) / BM, Nt = (g.N + BN - 1) / BN;
    int vid;
    {
        const int p = blockIdx.x, tot = Mt * Nt;
        const int q = tot / 8, r = tot % 8, xcd = p % 8, loc = p / 8;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // Which index runs fastest decides what stays hot in the XCD's L2 while the run is walked: M fastest keeps the weight
    // panel (right when W is the big operand: the small-M DiT products); N fastest keeps the A rows and re-streams the
    // (small, L2 / MALL resident) weights — right for the tall codec products, where M-fastest re-read all of A from HBM
    // once per N-tile (PMC: 896 MB per launch against 250 MB of operands at 24000 x 2048 x 512).
    // When BOTH operands are larger than an XCD's L2 (the codec's coarse ConvTranspose / FFN products) neither order is right:
    // N fastest re-streams the whole of W once per row tile (counters, 4800 x 2560 x 2048 at split-bf16: 615 MB per launch
    // against 110 MB of operands + output).  nfast = G > 1 walks bands of G row tiles, M fastest inside a band, so the ~32-64
    // tiles an XCD runs at once cover G row tiles x several column tiles and every operand k-slice is shared while it is hot.
    const int gm = g.nfast;
    int mi, ni;
    if (gm == 0) {
        mi = vid % Mt; ni = vid / Mt;
    } else {
        const int per = gm * Nt, band = vid / per, first = band * gm, loc = vid - band * per;
        const int rows = Mt - first < gm ? Mt - first : gm;
        mi = first + loc % rows; ni = loc / rows;
    }
    const int m0 = mi * BM, n0 = ni * BN, z = blockIdx.z;
    const int zb = g.ksplit_tiles ? 0 : z;  // batch index (split-K launches are unbatched)
    const long wz = (long)(g.w_zmod ? zb % g.w_zmod : zb) * g.w_z;

    // ---- per-lane DMA sources: slot -> (array, 8-row block) --------------------------------------
    const bf16_t* src[PW];
    unsigned dst[PW];  // wave-uniform LDS byte offset inside a stage
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int slot = wave * PW + i;  // wave-uniform
        const int rl = lane >> 3, p = lane & 7;
        if (DUMMY && slot >= NA + NWS) {
            int m = m0 + rl;
            m = m < g.M ? m : g.M - 1;
            src[i] = g.Ahi + (long)zb * g.a_z + g.amap.at(m) + p * 8;
            dst[i] = STAGE;
        } else if (slot < NA) {
            const int arr = slot / (BM / 8), rb = slot % (BM / 8);
            const int r = rb * 8 + rl;
            const int c = p ^ ((r >> 1) & 7);
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            src[i] = (arr ? g.Alo : g.Ahi) + (long)zb * g.a_z + g.amap.at(m) + c * 8;
            dst[i] = arr * A_ARR + rb * 1024;
        } else {
            const int s2 = slot - NA;
            const int arr = s2 / (BN / 8), rb = s2 % (BN / 8);
            const int r = rb * 8 + rl;
            const int c = p ^ ((r >> 1) & 7);
            int n = n0 + r;
            n = n < g.N ? n : g.N - 1;
            src[i] = (arr ? g.Wlo : g.Whi) + wz + (long)n * g.ldw + c * 8;
            dst[i] = NARR_A * A_ARR + arr * W_ARR + rb * 1024;
        }
    }

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
    const int nk_all = g.K / BK;
    const int kt0 = g.ksplit_tiles ? z * g.ksplit_tiles : 0;
    const int nk = g.ksplit_tiles ? (nk_all - kt0 < g.ksplit_tiles ? nk_all - kt0 : g.ksplit_tiles) : nk_all;
    const bf16_t* pf[PFN ? PFN : 1];
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
        int L = i * NW * 64 + tid;  // line index: [arrays of A rows | arrays of W rows]
        L = L < LINES ? L : LINES - 1;
        if (L < NARR * BM) {
            const int arr = L / BM;
            int m = m0 + L % BM;
            m = m < g.M ? m : g.M - 1;
            pf[i] = (arr ? g.Alo : g.Ahi) + (long)zb * g.a_z + g.amap.at(m);
        } else {
            const int L2 = L - NARR * BM, arr = L2 / BN;
            int n = n0 + L2 % BN;
            n = n < g.N ? n : g.N - 1;
            pf[i] = (arr ? g.Wlo : g.Whi) + wz + (long)n * g.ldw;
        }
    }
    // The touch is a 4-byte LDS-DMA into a 256-B dummy slot behind the ring: a VGPR destination would be written
    // asynchronously, long after hipcc may have reused that register (e.g. for a DMA address -> memory fault).
    auto touch4 = [&](const void* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dword %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    };
    auto issue = [&](int kt) {
        const unsigned st = lds0 + (unsigned)((kt % S) * STAGE_LD);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            dma16(src[i] + (kt0 + kt) * BK, st + (unsigned)__builtin_amdgcn_readfirstlane((int)dst[i]));
        int kp = kt0 + kt + PFD;
        kp = kp < nk_all ? kp : nk_all - 1;
#pragma unroll
        for (int i = 0; i < PFN; ++i)
            touch4(pf[i] + kp * BK, lds0 + (unsigned)(S * STAGE_LD));
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets inside a stage for the 4 k16-steps (constant over k-tiles)
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[TM][4], w_off[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a_off[i][kk] = r * 128 + (((kk * 2 + fh) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            w_off[j][kk] = NARR_A * A_ARR + r * 128 + (((kk * 2 + fh) ^ ((r >> 1) & 7)) << 4);
    }

    G3_STAMPR(152);
    G3_STAMP(0);
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s);
    G3_STAMP(1);
    if constexpr (epi_fold_in<Epi>::value) {
        // LN-fold consumer (gemm.hpp LnFoldIn): (mu, rstd) of the tile's rows from the producer's per-group partials, reduced in a
        // fixed order, into an LDS table behind the ring.  The loads are requested behind the ring's first stages and waited for with
        // them (the first k-tile waits for stage 0 anyway); the table is read in the epilogue only, many barriers later.
        float* const lstat = reinterpret_cast<float*>(smem + S * STAGE_LD + 256);
        epi.fold.lstat = lstat;
        epi.fold.m0 = m0;
        if (epi.fold.part && tid < BM) {
            int m = m0 + tid;
            m = m < g.M ? m : g.M - 1;
            const float4* p = reinterpret_cast<const float4*>(epi.fold.part + (long)m * epi.fold.NP * 2);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 5
            for (int i = 0; i < epi.fold.NP / 2; ++i) {
                const float4 v = p[i];
                s1 = (s1 + v.x) + v.z;
                s2 = (s2 + v.y) + v.w;
            }
            const float mu = epi.fold.rms ? 0.f : s1 * epi.fold.inv_c;
            const float var = fmaf(-mu, mu, s2 * epi.fold.inv_c);
            lstat[2 * tid] = mu;
            lstat[2 * tid + 1] = 1.0f / sqrtf((var > 0.f ? var : 0.f) + epi.fold.eps);
        }
    }

    for (int kt = 0; kt < nk; ++kt) {
        // younger than stage kt's DMA: its own touch loads + (S-2) full later stages
        if (kt + S - 1 <= nk)
            wait_vmcnt<(S - 2) * OPS + PFN>();
        else
            wait_vmcnt<0>();
        G3_STAMPK(2 + 4 * kt);          // own DMA pieces of k-tile kt landed
        __builtin_amdgcn_s_barrier();
        G3_STAMPK(3 + 4 * kt);          // everybody's did
        const char* st = smem + (kt % S) * STAGE_LD;
        // Small latency-bound tiles (64x64: the DiT's N = 960 projections) issue every fragment read of the k-tile first, THEN the
        // next stage's DMAs (4 wave-instructions of ~60 cycles each, M0 dance included), then the MFMAs: the LDS latency of the
        // reads hides under the DMA issue instead of following it — per-k-tile timeline (tools/gemm3_timeline.py,
        // profiles/r03n_*, r03o_*): wait 31 % / barrier 9 % / DMA issue 20 % / reads + MFMA 40 % of 0.77 us before, -7 % after.
        // The larger tiles keep the interleaved order: with 8 - 32 MFMAs per k-tile the compiler already hides the reads under
        // them, and holding all fragments delays the first MFMA (24000 x 512 x 2048: 74 -> 87 us when forced, r03o).
#ifndef G3_LATE_64
#define G3_LATE_64 1   // (A/B: -DG3_LATE_64=0 restores the interleaved order everywhere)
#endif
        constexpr bool LATE = G3_LATE_64 && BM == 64 && BN == 64;
        if constexpr (LATE) {
            bf16x8 ah[4][TM], al[4][SPLIT == 3 ? TM : 1], bh[4][TN], bl[4][NARR_W == 2 ? TN : 1];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[kk][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i][kk]);
                    if (SPLIT == 3) al[kk][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i][kk] + A_ARR);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[kk][j] = *reinterpret_cast<const bf16x8*>(st + w_off[j][kk]);
                    if (NARR_W == 2) bl[kk][j] = *reinterpret_cast<const bf16x8*>(st + w_off[j][kk] + W_ARR);
                }
            }
            if (kt + S - 1 < nk) issue(kt + S - 1);
            G3_STAMPK(4 + 4 * kt);          // fragment reads + next stage's DMAs issued
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (SPLIT == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk][i], bh[kk][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk][i], bl[kk][j], acc[i][j], 0, 0, 0);
                        }
                        if (SPLIT == PREC_F16X2) acc[i][j] = mfma16<SPLIT>(ah[kk][i], bl[kk][j], acc[i][j]);
                        acc[i][j] = mfma16<SPLIT>(ah[kk][i], bh[kk][j], acc[i][j]);
                    }
        } else {
            if (kt + S - 1 < nk) issue(kt + S - 1);
            G3_STAMPK(4 + 4 * kt);          // next stage's DMAs issued
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i][kk]);
                    if (SPLIT == 3) al[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i][kk] + A_ARR);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(st + w_off[j][kk]);
                    if (NARR_W == 2) bl[j] = *reinterpret_cast<const bf16x8*>(st + w_off[j][kk] + W_ARR);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (SPLIT == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        if (SPLIT == PREC_F16X2) acc[i][j] = mfma16<SPLIT>(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = mfma16<SPLIT>(ah[i], bh[j], acc[i][j]);
                    }
            }
        }
        G3_STAMP_FENCE(acc[0][0][0]);   // (lab: the stamp must not move above the MFMAs' issue)
        G3_STAMPK(5 + 4 * kt);          // fragment reads + MFMAs of k-tile kt issued
    }
    G3_STAMP(150);
    if constexpr (Epi::TILE) {   // whole-workgroup epilogue through an fp32 LDS tile in the finished ring (EpiQKV)
        static_assert(BN == 128 && (BM == 64 || BM == 128), "tile epilogue: 128-column tiles of 64 / 128 rows");
        static_assert(64 * Epi::TP * 4 <= S * STAGE_LD, "tile epilogue: the fp32 tile must fit the finished ring");
        epi.template tile_epilogue<BM, TM, TN, WN, NW>(acc, g.M, m0, n0, wave, lane, reinterpret_cast<float*>(smem));
        G3_STAMP(151);
        G3_STAMPR(153);
        return;
    }
    if constexpr (Epi::STAGE16) {
        constexpr int W = Epi::PAIRED ? 32 : TN * 32;
        constexpr int STG = 32 * (2 * W + 16);   // bytes of one wave's staging tile
        static_assert(NW * STG <= S * STAGE_LD, "staging tiles must fit the finished ring");
        if (g.stage16 && epi.stage16_ok() && ((Epi::PAIRED ? g.N / 2 : g.N) % 8) == 0) {   // wave-uniform: kernel arguments only
            __syncthreads();   // every wave has read its last fragments: the ring is free
            gemm_epilogue_staged16<TM, TN, Epi>(epi, acc, g.M, g.N, m0 + wm * TM * 32, n0 + wn * TN * 32, z, lane, smem + wave * STG);
            G3_STAMP(151);
            G3_STAMPR(153);
            return;
        }
    }
    gemm_epilogue<TM, TN, Epi>(epi, acc, g.M, g.N, m0 + wm * TM * 32, n0 + wn * TN * 32, z, lane);
    G3_STAMP(151);
    G3_STAMPR(153);
}

template <int BM, int BN, int WM, int WN, int SPLIT, int S, class Epi>
static inline hipError_t gemm3_launch_cfg(const Gemm3Operands& g, const Epi& epi, int Z, hipStream_t st) {
    constexpr int NARR_A = SPLIT == 3 ? 2 : 1, NARR_W = (SPLIT == 3 || SPLIT == PREC_F16X2) ? 2 : 1;
    constexpr bool DUMMY = (NARR_A * (BM / 8) + NARR_W * (BN / 8)) % (WM * WN) != 0;
    constexpr size_t lds = (size_t)S * ((NARR_A * BM + NARR_W * BN) * 128 + (DUMMY ? 1024 : 0)) + 256  // ring (+ pads) + dummy slot of the prefetch touches
                           + (epi_fold_in<Epi>::value ? BM * 8 : 0);                                    // + (mu, rstd) of the tile's rows (LN-fold consumers)
    static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, Z);  // 1-D tile index, remapped per XCD in-kernel
    auto kern = gemm3_kernel<BM, BN, WM, WN, SPLIT, S, Epi>;
    static DevOnce once;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, g, epi);
    return hipGetLastError();
}

enum Gemm3Cfg {
    G3_64x128 = 0,   // 8 waves 2x4, wave 32x32, 3 stages (144 KiB): general
    G3_128x128 = 1,  // 8 waves 4x2, wave 32x64, 2 stages (128 KiB): SwiGLU pairs, large M x N
    G3_64x64 = 2,    // 4 waves 2x2, wave 32x32, 2 stages (64 KiB) -> 2 workgroups / CU: small N
    G3_128x64 = 3,   // 8 waves 4x2, wave 32x32, 3 stages (144 KiB): tall, N <= 64
    G3_128x32 = 4,   // 4 waves 4x1, wave 32x32, 2 stages (80 KiB) -> 2 workgroups / CU: tall, N <= 32
    G3_160x128 = 5,  // 10 waves 5x2, wave 32x64, 2 stages (146 KiB): M = 600 (4 row tiles, 6 % padding) x wide N in ONE round
    G3_128x128_W4 = 6,  // 4 waves 2x2, wave 64x64: 4 fragment reads per 4 MFMAs instead of 3 per 2 — for the single-array formats,
                        // where one MFMA per fragment pair leaves the k-loop bound by ds_read_b128 traffic; 2 workgroups per CU
    G3_64x32 = 8,       // 2 waves 2x1, wave 32x32 (12 KiB per single-array stage): twice the workgroups of 64x64 on the N = 960 projections (300 at M = 600)
    G3_32x64 = 9,       // 2 waves 1x2
    G4_256x256 = 7,     // gemm4.hpp: 8 waves 2x4, wave 128x64, two 64-KiB k-tile buffers on the phase-split schedule (single-array formats)
};
static inline bool gemm3_ok(const Gemm3Operands& g) {
    return g.K % 64 == 0 && g.K >= 64 && (g.amap.ld % 8) == 0 && (g.amap.off % 8) == 0 && (g.ldw % 8) == 0 &&
           (g.amap.bstride % 8) == 0;
}
// gemm4 (256 x 256 x 64 macro-tile on a phase-split schedule, round 5): bit-identical to gemm3 and 1.05 PFLOP/s on 4096^3, but no better
// than gemm3 on any product this library runs (profiles/r05b-d, r05h) — it lives in exp/ and is compiled into lab builds only
// (make LAB=1, cfg 7; tools/gemm4_check.py); the shipped library has no instantiation of it.
#ifdef SMTTS_LAB
#include "exp/gemm4.hpp"
#else
#include <type_traits>
template <class Epi> struct gemm4_enabled : std::false_type {};
static inline bool gemm4_ok(const Gemm3Operands&) { return false; }
#endif

static inline int gemm3_pick_cfg(int M, int N, bool paired, bool single = false /* one array per operand (fp16 / bf16) */) {
    extern int g_gemm3_w4_minm;   // single-array formats: 128x128 with four 64x64 waves from this M up (0 = never); split-bf16 falls back
    // round 2 (profiles/r02k_ab_keepx_w4.txt) took the 4-wave tile for the wide, short-K first FFN product of the codec's GEMM stages
    // (24000 x 2048 x 512: 122 -> 98 us); with the LDS-staged 16-bit epilogue the 8-wave 128x128 tile now beats it on those very
    // shapes (round 3, profiles/r03j_gemm_codec_tile_sweep.txt: 101 vs 90 us, 4800 x 4096 x 1024: 69 vs 62 us) -> off by default
    // (SMTTS_GEMM_W4_MINM=<M> switches it back on); 256x256 and 256x128 tiles were measured there too and bought nothing.
    if (g_gemm3_w4_minm > 0 && M >= g_gemm3_w4_minm && N >= 2048) return G3_128x128_W4;
    if (single && !paired && M >= 2048 && N >= 2048) return G3_128x128;
    if (paired) {
        // the SwiGLU pair epilogue needs 32x64 wave tiles: 128x128 or 160x128.  Single-array formats run two such workgroups per
        // CU, so 512 tiles are one round: the teacher's 1800 x 4800 FF1 is 570 tiles of 128x128 (two rounds) but 456 of 160x128
        const long t128p = (long)((M + 127) / 128) * ((N + 127) / 128), t160p = (long)((M + 159) / 160) * ((N + 127) / 128);
        if (single && t128p > 512 && t160p <= 512) return G3_160x128;
        return G3_128x128;
    }
    if (N <= 32) return G3_128x32;
    if (N <= 64) return M >= 2048 ? G3_128x64 : G3_64x64;
    // cost model measured on MI355X: time ~ rounds(tiles / 256 CUs) x bytes ingested per workgroup / ~40 GB/s.
    // 128x128 moves the fewest bytes per flop; prefer it whenever it fills at least half the chip in ONE round or
    // many rounds (a 64x128 grid of 257..511 tiles costs two rounds, e.g. DiT QKVG: 300 tiles 36.5 us vs 150 tiles 29.5 us)
    extern int g_gemm3_t160;
    const long t160 = (long)((M + 159) / 160) * ((N + 127) / 128);
    if (g_gemm3_t160 && M <= 640 && M > 480 && t160 > 128 && t160 <= 256) return G3_160x128;
    if (g_gemm3_t160 && M > 640 && N >= 128) {
        // rounds x measured time of one k-tile round (us: 64x128 1.03, 128x128 2.0, 160x128 2.37; one workgroup per CU each):
        // the 160-row tile wins where it saves a round, e.g. 4800 x 1024: 600 tiles of 64x128 = 3 rounds vs 240 = 1 round
        auto rounds = [](long t) { return (double)((t + 255) / 256); };
        const double c64 = rounds((long)((M + 63) / 64) * ((N + 127) / 128)) * 1.03;
        const double c128 = rounds((long)((M + 127) / 128) * ((N + 127) / 128)) * 2.0;
        const double c160 = rounds(t160) * 2.37;
        if (c160 < 0.96 * c64 && c160 < 0.96 * c128) return G3_160x128;
    }
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (single) {
        // single-array formats hold two 128x128 workgroups (2 x 64 KiB of ring) or four 64x64 ones per CU, so their one-round
        // capacities are 512 / 1024 tiles (profiles/r02ae_gemm_cfg_sweep.txt, fp16): the teacher's 1800 x 3840 x 960 QKVG as 450
        // tiles of 128x128 27.5 us against 34.4 as 870 of 64x128; the B = 8 QKVG (600 x 3840) as 600 tiles of 64x64 15.3 against 17.6
        const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
        if (M <= 640 && t128 >= 128 && t128 <= 256 && t64 > 512 && t64 <= 1024) return G3_64x64;
        if (t128 > 256 && t128 < 512) return G3_128x128;
    }
    if (t128 >= 512 || (t128 >= 128 && t128 <= 256)) return G3_128x128;
    const long t64x128 = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (t64x128 >= 200) return G3_64x128;
    return G3_64x64;
}


// Ring depth per tile shape.  A single-array operand format (PREC_F16 / PREC_BF16) halves the bytes per stage, so the same LDS
// budget can hold twice the k-tiles in flight.  Measured (profiles/r02b_ab_ring_depth.txt, f16, M = 600): deep rings cut one
// batch's latency (QKVG 25.4 -> 21.5 us, SwiGLU 22.7 -> 19.4, 64x64 16.0 -> 14.5; 14.8 -> 14.2 ms per batch one at a time)
// but cost throughput with several batches in flight (10.1 -> 10.4 ms per batch): a workgroup that holds 128 KiB of LDS while
// it waits on memory keeps the other streams' kernels off its CU.  Hence a run-time choice: g_gemm3_deep (Engine tuning mode).
template <int SPLIT, class Epi>
static inline hipError_t gemm3_launch_split(const Gemm3Operands& g, const Epi& epi, int Z, int cfg, hipStream_t st) {
    extern thread_local int g_gemm3_deep;
    if constexpr (SPLIT == PREC_BF16 || SPLIT == PREC_F16) {
        // deep rings pay when the whole grid is resident at once (one latency-bound round); a grid of several rounds at the deep
        // ring's occupancy runs faster shallow with more workgroups per CU (teacher QKVG, 450 tiles of 128x128: 35.1 us deep — two
        // rounds at one workgroup per CU — against 27.5 shallow; B = 8 QKVG as 600 tiles of 64x64: 20.4 against 15.3)
        const long bm = cfg == G4_256x256 ? 256 : cfg == G3_64x128 || cfg == G3_64x64 || cfg == G3_64x32 ? 64 : cfg == G3_32x64 ? 32 : cfg == G3_160x128 ? 160 : 128,
                   bn = cfg == G4_256x256 ? 256 : cfg == G3_64x64 || cfg == G3_32x64 ? 64 : cfg == G3_64x32 ? 32 : 128;
        const long tiles = ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * (Z > 0 ? Z : 1);
        const long deep_slots = cfg == G3_64x64 ? 512 : 256;
        if (g_gemm3_deep && tiles <= deep_slots) {
            switch (cfg) {
                case G3_128x128:
                    return gemm3_launch_cfg<128, 128, 4, 2, SPLIT, 4, Epi>(g, epi, Z, st);
                case G3_64x128:
                    if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<64, 128, 2, 4, SPLIT, 6, Epi>(g, epi, Z, st);
                    break;
                case G3_64x64:
                    if constexpr (epi_small_n<Epi>::value) {   // the DiT's N = 960 residual projections: 150 workgroups, one per CU — the LDS is there for a longer ring
                        if (g_gemm3_deep >= 2) return gemm3_launch_cfg<64, 64, 2, 2, SPLIT, 8, Epi>(g, epi, Z, st);
                    }
                    if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<64, 64, 2, 2, SPLIT, 4, Epi>(g, epi, Z, st);
                    break;
                case G3_160x128:
                    return gemm3_launch_cfg<160, 128, 5, 2, SPLIT, 4, Epi>(g, epi, Z, st);
                default:
                    break;  // the tall 128x64 / 128x32 shapes (codec, M >= 2048) keep their depth: many rounds, never latency-bound
            }
        }
    }
    if constexpr (SPLIT == 3 && !Epi::PAIRED) {
        // split-bf16 on the 64 x 64 tile: four arrays per stage, so the 2-stage ring has ONE k-tile in flight and every k-tile pays a
        // whole memory round trip (the grouped conv pos-embed: 31 k-tiles, 1.26 us each).  A grid that is resident at once (<= 256
        // workgroups, one per CU: 128 KiB of LDS) takes four stages (round 6)
        const long tiles64 = ((g.M + 63) / 64) * ((g.N + 63) / 64) * (Z > 0 ? Z : 1);
        if (g_gemm3_deep && cfg == G3_64x64 && tiles64 <= 256) return gemm3_launch_cfg<64, 64, 2, 2, 3, 4, Epi>(g, epi, Z, st);
    }
    switch (cfg) {
#ifdef SMTTS_LAB
        case G4_256x256:
            if constexpr (SPLIT == PREC_BF16 || SPLIT == PREC_F16) return gemm4_launch_cfg<SPLIT, Epi>(g, epi, Z, st);
            break;
#endif
        case G3_128x128:
            return gemm3_launch_cfg<128, 128, 4, 2, SPLIT, 2, Epi>(g, epi, Z, st);
        case G3_64x128:
            if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<64, 128, 2, 4, SPLIT, 3, Epi>(g, epi, Z, st);
            break;
        case G3_64x64:
            if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<64, 64, 2, 2, SPLIT, 2, Epi>(g, epi, Z, st);
            break;
        case G3_128x64:
            if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<128, 64, 4, 2, SPLIT, 3, Epi>(g, epi, Z, st);
            break;
        case G3_128x32:
            if constexpr (!Epi::PAIRED) return gemm3_launch_cfg<128, 32, 4, 1, SPLIT, 2, Epi>(g, epi, Z, st);
            break;
        case G3_160x128:   // (32x64 wave tiles: also valid for the paired SwiGLU epilogue)
            return gemm3_launch_cfg<160, 128, 5, 2, SPLIT, 2, Epi>(g, epi, Z, st);
        case G3_128x128_W4:
            if constexpr (SPLIT == PREC_BF16 || SPLIT == PREC_F16) return gemm3_launch_cfg<128, 128, 2, 2, SPLIT, 2, Epi>(g, epi, Z, st);
            break;
        case G3_64x32:
            if constexpr (epi_small_n<Epi>::value && SPLIT != 3) return g_gemm3_deep >= 2 ? gemm3_launch_cfg<64, 32, 2, 1, SPLIT, 8, Epi>(g, epi, Z, st)
                                                                                         : gemm3_launch_cfg<64, 32, 2, 1, SPLIT, 4, Epi>(g, epi, Z, st);
            break;
        case G3_32x64:
            if constexpr (epi_small_n<Epi>::value && SPLIT != 3) return g_gemm3_deep >= 2 ? gemm3_launch_cfg<32, 64, 1, 2, SPLIT, 8, Epi>(g, epi, Z, st)
                                                                                         : gemm3_launch_cfg<32, 64, 1, 2, SPLIT, 4, Epi>(g, epi, Z, st);
            break;
    }
    return hipErrorInvalidValue;
}

template <class Epi>
static inline hipError_t gemm3_launch(const Gemm3Operands& g_in, const Epi& epi, int Z, int split, hipStream_t st,
                                      int cfg = -1) {
    const Gemm3Operands& g0 = g_in;
    if (g0.M <= 0 || g0.N <= 0) return hipSuccess;
    if (!gemm3_ok(g0)) return hipErrorInvalidValue;
    if (cfg < 0) cfg = gemm3_pick_cfg(g0.M, g0.N, Epi::PAIRED, split != PREC_BF16X3);
    extern int g_gemm3_nfast;
    Gemm3Operands g = g_in;
    extern int g_gemm3_stage16;
    g.stage16 = g_gemm3_stage16;
    // the bigger operand streams, the smaller stays in L2 — unless the smaller one does not fit there either (> 3 MB): bands
    extern int g_gemm3_group;
    const double wbytes = (double)g.N * g.K * (split == PREC_BF16X3 ? 4.0 : 2.0);
    g.nfast = g_gemm3_nfast && (long)g.M > (long)g.N ? (g_gemm3_group > 1 && wbytes > 3e6 && g.M >= 1024 ? g_gemm3_group : 1) : 0;
    if ((cfg == G3_64x32 || cfg == G3_32x64) && (split == PREC_BF16X3 || !epi_small_n<Epi>::value)) cfg = G3_64x64;   // (no such instantiation)
    if (cfg == G4_256x256 && (split == PREC_BF16X3 || !gemm4_ok(g) || !gemm4_enabled<Epi>::value)) cfg = G3_128x128;   // (no such instantiation)
    if (split == PREC_BF16X3) return gemm3_launch_split<3, Epi>(g, epi, Z, cfg == G3_128x128_W4 ? G3_128x128 : cfg, st);
    if (split == PREC_F16) return gemm3_launch_split<2, Epi>(g, epi, Z, cfg, st);
    return gemm3_launch_split<1, Epi>(g, epi, Z, cfg, st);
}

// PREC_F16X2: one fp16 array of A (g.Ahi) against an fp16 hi + lo pair of W (g.Whi, g.Wlo) — two MFMA passes instead of the
// three of split-bf16, A rounded to 11 bits, W exact to ~22.  Instantiated only where it is used (the codec's ConvTranspose
// products, gemm3_store.hip): not part of gemm3_launch's run-time precision switch.
template <class Epi>
static inline hipError_t gemm3_launch_x2(const Gemm3Operands& g_in, const Epi& epi, int Z, hipStream_t st, int cfg = -1) {
    if (g_in.M <= 0 || g_in.N <= 0) return hipSuccess;
    if (!gemm3_ok(g_in) || !g_in.Wlo) return hipErrorInvalidValue;
    if (cfg < 0) cfg = gemm3_pick_cfg(g_in.M, g_in.N, Epi::PAIRED, false);
    extern int g_gemm3_nfast, g_gemm3_stage16, g_gemm3_group;
    Gemm3Operands g = g_in;
    g.stage16 = g_gemm3_stage16;
    const double wbytes = (double)g.N * g.K * 4.0;
    g.nfast = g_gemm3_nfast && (long)g.M > (long)g.N ? (g_gemm3_group > 1 && wbytes > 3e6 && g.M >= 1024 ? g_gemm3_group : 1) : 0;
    return gemm3_launch_split<PREC_F16X2, Epi>(g, epi, Z, cfg == G3_128x128_W4 ? G3_128x128 : cfg, st);
}
