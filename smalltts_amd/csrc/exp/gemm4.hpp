// GEMM v4 for gfx950 — 256 x 256 x 64 macro-tile on a phase-split, deep-pipelined schedule (round 5).
//   C[M,N] = A[M,K] x W[N,K]^T, both operands ONE 16-bit array each in HBM (PREC_F16 / PREC_BF16), fp32 accumulate.
//
// Why another generation.  gemm3's ring (one barrier + one full-stage vmcnt wait per k-tile, every wave doing
// wait -> barrier -> DMA issue -> fragment reads -> MFMAs in lock-step) tops out at 525-611 TFLOP/s on the codec's wide
// products (profiles/r04z_phases.txt): per CU it ingests ~33-40 GB/s from L2 whatever the tile, so a 128 x 128 tile's
// 2.1 MFLOP per 32 KiB k-tile is the ceiling, and a 256 x 256 tile on the SAME ring bought nothing (r03j: one workgroup per
// CU, nothing under the wait).  cdna_hip_programming.md §5 "The 256^2 8-phase template": twice the flops per ingested byte
// only pays on a schedule whose loads stay in flight ACROSS barriers and whose two waves per SIMD alternate roles.
//
// Structure (8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32, 128 accumulator registers):
// * LDS: two k-tile buffers of [A 256 rows | W 256 rows] x 128 B = 64 KiB each, rows XOR-swizzled on the DMA SOURCE address
//   exactly like gemm3 (16-B chunk c of row r at position c ^ ((r >> 1) & 7): conflict-free ds_read_b128, measured there).
// * A k-tile is staged as four 16-KiB UNITS that follow the order in which a k-tile is read:
//     U0 = A rows {h*128 + [0,64)}   U1 = W rows {q*64 + [0,32)}   U2 = W rows {q*64 + [32,64)}   U3 = A rows {h*128 + [64,128)}
//   (h = wave row 0 / 1, q = wave column 0..3): every wave issues two 1-KiB global_load_lds_dwordx4 pieces per unit.
// * A k-tile is computed in four PHASES of 8 MFMAs (one 64 x 32 quadrant of the wave tile x K = 64) each:
//     phase 0: reads A rows 0-63 (8 ds_read_b128) + W cols 0-31 (4)   -> tiles (0..1, 0)
//     phase 1: reads W cols 32-63 (4)                                   -> tiles (0..1, 1)
//     phase 2: reads A rows 64-127 (8)                                  -> tiles (2..3, 1)
//     phase 3: nothing (W cols 0-31 stayed in registers)                -> tiles (2..3, 0)
//   phase = [fragment reads | issue ONE unit | s_waitcnt vmcnt(8)] s_barrier [8 MFMAs at s_setprio 1] s_barrier.
//   The unit issued in a phase is needed five phases later (U2 / U3 of the next k-tile in phases 0 / 1, U0 / U1 of the one
//   after in phases 2 / 3, into the slots whose last read was >= 2 phases ago), so FOUR units = 8 DMA instructions per wave
//   stay in flight across every barrier: vmcnt is never 0 inside the loop.  The wait of phase p retires the unit read in
//   phase p + 1: two barriers lie between a wave's wait and anybody's read of that data (one more than needed unstaggered,
//   which the stagger below consumes).
// * The wave rows are STAGGERED by one barrier (wave row 1 executes one extra s_barrier up front, wave row 0 one at the end):
//   each SIMD holds one wave of either row, so while one issues reads / DMAs the other runs its MFMAs — the role split that
//   s_setprio needs to arbitrate (guide T3 + T5).
// Requires K % 64 == 0, K >= 128, 16-B aligned rows (gemm3_ok).  Epilogue functors and the accumulator layout are gemm3's.
// (included by gemm3.hpp in front of its launchers: never include this file directly)
#pragma once
#include <type_traits>

#ifndef G4_SETPRIO
#define G4_SETPRIO 1
#endif
#ifndef G4_STAGGER
#define G4_STAGGER 1
#endif
#ifndef G4_WAITN      // timing experiments only: steady-state vmcnt (8 = four units in flight; < 8 still correct, > 8 NOT)
#define G4_WAITN 8
#endif
#ifndef G4_NODMA      // timing experiment only (wrong results): no unit DMAs inside the main loop
#define G4_NODMA 0
#endif

#ifdef G4_TIMELINE   // debug build (tools/gemm4_timeline.py): waves 0 and 4 of the first 256 workgroups stamp the shader clock four times per phase
static __device__ unsigned long long g4_tl_buf[256 * 2 * 2 * 16 + 8];   // [workgroup][wave row][k-tile 4 / 5][phase][start | waited | barrier 1 + reads | MFMAs issued]
#define G4_STAMP(i) do { if (tl_on && lane == 0) g4_tl_buf[tl_base + P * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G4_STAMP(i) do { } while (0)
#endif

template <int SPLIT, class Epi>
__global__ __launch_bounds__(512) void gemm4_kernel(Gemm3Operands g, Epi epi) {
    static_assert(SPLIT == PREC_F16 || SPLIT == PREC_BF16, "gemm4: single-array operand formats only");
    constexpr int BM = 256, BN = 256, BK = 64, WN = 4, TM = 4, TN = 2;
    constexpr int A_BYTES = BM * 128, BUF = (BM + BN) * 128;   // 32 KiB of A rows, then 32 KiB of W rows, per k-tile buffer
    static_assert(!Epi::PAIRED && !Epi::TILE, "gemm4: column epilogues only");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware tile order: as gemm3 (every XCD a contiguous run of virtual tile ids; bands of row tiles when nfast > 1)
    const int Mt = (g.M + BM - 1) / BM, Nt = (g.N + BN - 1) / BN;
    int vid;
    {
        const int p = blockIdx.x, tot = Mt * Nt;
        const int q = tot / 8, r = tot % 8, xcd = p % 8, loc = p / 8;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
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
    const int nk_all = g.K / BK;
    const int kt0 = g.ksplit_tiles ? z * g.ksplit_tiles : 0;
    const int nk = g.ksplit_tiles ? (nk_all - kt0 < g.ksplit_tiles ? nk_all - kt0 : g.ksplit_tiles) : nk_all;
    const int zb = g.ksplit_tiles ? 0 : z;
    const long wz = (long)(g.w_zmod ? zb % g.w_zmod : zb) * g.w_z;

    // ---- DMA sources: unit u, piece i of this wave (pieces 2 wave, 2 wave + 1 of the unit's sixteen) ----------------------
    const bf16_t* src[4][2];
    unsigned dst[4][2];   // LDS byte offset of the piece inside a buffer (wave-uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave * 2 + i;                 // piece 0..15 of the unit
            const bool isA = (u == 0 || u == 3);
            const int r0 = isA ? (j >> 3) * 128 + (u == 3 ? 64 : 0) + (j & 7) * 8
                               : (j >> 2) * 64 + (u == 2 ? 32 : 0) + (j & 3) * 8;
            const int r = r0 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            if (isA) {
                int m = m0 + r;
                m = m < g.M ? m : g.M - 1;
                src[u][i] = g.Ahi + (long)zb * g.a_z + g.amap.at(m) + c * 8 + (long)kt0 * BK;
                dst[u][i] = (unsigned)(r0 * 128);
            } else {
                int n = n0 + r;
                n = n < g.N ? n : g.N - 1;
                src[u][i] = g.Whi + wz + (long)n * g.ldw + c * 8 + (long)kt0 * BK;
                dst[u][i] = (unsigned)(A_BYTES + r0 * 128);
            }
        }
    auto dma16 = [&](const void* gsrc, unsigned lds_dst) __attribute__((always_inline)) {
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
    auto issue = [&](int u, int kt) __attribute__((always_inline)) {   // unit u of k-tile kt (u compile-time after unrolling, kt run-time)
        const unsigned st = lds0 + (unsigned)((kt & 1) * BUF);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma16(src[u][i] + (long)kt * BK, st + (unsigned)__builtin_amdgcn_readfirstlane((int)dst[u][i]));
    };

    // eight named accumulators (a [4][2] array captured by the phase lambda stayed an alloca: spilled to scratch between the
    // loop and the tail phases); c<i><j> = MFMA tile (i, j) of the wave tile
    floatx16 c00, c01, c10, c11, c20, c21, c30, c31;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; c20[r] = 0.f; c21[r] = 0.f; c30[r] = 0.f; c31[r] = 0.f; }

    // fragment byte offsets inside a buffer for the 4 k16 steps: row fr of the wave's first A / W tile (+ i * 4096 per 32 rows)
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[4], w_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int ch = ((kk * 2 + fh) ^ ((fr >> 1) & 7)) << 4;   // (rows i * 32 + wave offsets are multiples of 16: same swizzle)
        a_off[kk] = (wm * 128 + fr) * 128 + ch;
        w_off[kk] = A_BYTES + (wn * 64 + fr) * 128 + ch;
    }

    // ---- prologue: k-tile 0 whole, U0 / U1 of k-tile 1 ---------------------------------------------------------------------
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    issue(0, 1); issue(1, 1);
    wait_vmcnt<8>();                      // own pieces of U0(0), U1(0) landed
    __builtin_amdgcn_s_barrier();         // ... everybody's did
    if (G4_STAGGER && wm == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind from here on

    bf16x8 af[2][4], b0f[4], b1f[4];
#ifdef G4_TIMELINE
    bool tl_on = false;
    int tl_base = 0;
#endif
    // One phase.  P = 0..3; ISSUE: unit to stage (-1 none) of k-tile `ikt`; WAIT: vmcnt to wait down to (-1 none)
    auto phase = [&](auto Pc, auto Uc, auto Wc, const char* buf, int ikt, floatx16& ca, floatx16& cb) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, U = decltype(Uc)::value, W = decltype(Wc)::value;
        G4_STAMP(0);
        if constexpr (P == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b0f[kk] = *reinterpret_cast<const bf16x8*>(buf + w_off[kk]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) af[i][kk] = *reinterpret_cast<const bf16x8*>(buf + a_off[kk] + i * 4096);
        } else if constexpr (P == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b1f[kk] = *reinterpret_cast<const bf16x8*>(buf + w_off[kk] + 4096);
        } else if constexpr (P == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) af[i][kk] = *reinterpret_cast<const bf16x8*>(buf + a_off[kk] + (2 + i) * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (U >= 0 && !(G4_NODMA && W == 8)) issue(U, ikt);
        if constexpr (W >= 0) wait_vmcnt<(W == 8 ? G4_WAITN : W)>();
        __builtin_amdgcn_sched_barrier(0);
        G4_STAMP(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G4_STAMP(2);
        if (G4_SETPRIO) __builtin_amdgcn_s_setprio(1);
        constexpr int J = (P == 0 || P == 3) ? 0 : 1;   // ca / cb = tiles (I0, J), (I0 + 1, J), I0 = P < 2 ? 0 : 2
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            ca = mfma16<SPLIT>(af[0][kk], J ? b1f[kk] : b0f[kk], ca);
            cb = mfma16<SPLIT>(af[1][kk], J ? b1f[kk] : b0f[kk], cb);
        }
        if (G4_SETPRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef G4_TIMELINE
        asm volatile("s_nop 0" :: "v"(ca[0]), "v"(cb[0]));   // (the stamp must not move above the MFMAs' issue)
#endif
        G4_STAMP(3);
        __builtin_amdgcn_s_barrier();
    };
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>;
    using I2_ = std::integral_constant<int, 2>; using I3_ = std::integral_constant<int, 3>;
    using I4_ = std::integral_constant<int, 4>; using I8_ = std::integral_constant<int, 8>;
    using N_ = std::integral_constant<int, -1>;

    int kt = 0;
#ifdef G4_TIMELINE
    if (tid == 0 && blockIdx.x == 0) g4_tl_buf[256 * 2 * 2 * 16] = __builtin_amdgcn_s_memrealtime(), g4_tl_buf[256 * 2 * 2 * 16 + 1] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (; kt < nk - 2; ++kt) {           // steady state: every phase stages one unit, eight DMAs stay in flight
        const char* buf = smem + (kt & 1) * BUF;
#ifdef G4_TIMELINE
        tl_on = (kt == 4 || kt == 5) && blockIdx.x < 256 && (wave & 3) == 0;
        tl_base = ((blockIdx.x * 2 + (wave >> 2)) * 2 + (kt - 4)) * 16;
#endif
        phase(I0_{}, I2_{}, I8_{}, buf, kt + 1, c00, c10);
        phase(I1_{}, I3_{}, I8_{}, buf, kt + 1, c01, c11);
        phase(I2_{}, I0_{}, I8_{}, buf, kt + 2, c21, c31);
        phase(I3_{}, I1_{}, I8_{}, buf, kt + 2, c20, c30);
    }
#ifdef G4_TIMELINE
    tl_on = false;
    if (tid == 0 && blockIdx.x == 0) g4_tl_buf[256 * 2 * 2 * 16 + 2] = __builtin_amdgcn_s_memrealtime(), g4_tl_buf[256 * 2 * 2 * 16 + 3] = __builtin_amdgcn_s_memtime();
#endif
    {   // second-to-last k-tile: the last k-tile's U2 / U3 go out, nothing beyond
        const char* buf = smem + (kt & 1) * BUF;
        phase(I0_{}, I2_{}, I8_{}, buf, kt + 1, c00, c10);
        phase(I1_{}, I3_{}, I8_{}, buf, kt + 1, c01, c11);
        phase(I2_{}, N_{}, N_{}, buf, 0, c21, c31);
        phase(I3_{}, N_{}, I4_{}, buf, 0, c20, c30);     // U0 / U1 of the last k-tile landed; its U2 / U3 may still fly
        ++kt;
    }
    {   // last k-tile
        const char* buf = smem + (kt & 1) * BUF;
        phase(I0_{}, N_{}, I2_{}, buf, 0, c00, c10);
        phase(I1_{}, N_{}, I0_{}, buf, 0, c01, c11);
        phase(I2_{}, N_{}, N_{}, buf, 0, c21, c31);
        phase(I3_{}, N_{}, N_{}, buf, 0, c20, c30);
    }
    if (G4_STAGGER && wm == 0) __builtin_amdgcn_s_barrier();   // pairs with wave row 1's last barrier: both rows aligned again
    // every wave is past its last fragment read and no DMA is in flight: the ring is free for the staged epilogue
    // write-out one 32-row tile at a time (TM = 1 calls): hipcc does not unroll the shared epilogues' row-tile loop at TM = 4 and
    // would index the accumulators dynamically, i.e. through scratch
    const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
    bool staged = false;
    if constexpr (Epi::STAGE16) staged = g.stage16 && epi.stage16_ok() && (g.N % 8) == 0;   // wave-uniform: kernel arguments only
    auto out_tile = [&](int i, const floatx16& t0, const floatx16& t1) __attribute__((always_inline)) {
        floatx16 t[1][TN] = {{t0, t1}};
        if constexpr (Epi::STAGE16) {
            constexpr int STG = 32 * (2 * TN * 32 + 16);   // bytes of one wave's staging tile
            if (staged) {
                gemm_epilogue_staged16<1, TN, Epi>(epi, t, g.M, g.N, mw0 + i * 32, nw0, z, lane, smem + wave * STG);
                return;
            }
        }
        gemm_epilogue<1, TN, Epi>(epi, t, g.M, g.N, mw0 + i * 32, nw0, z, lane);
    };
    out_tile(0, c00, c01);
    out_tile(1, c10, c11);
    out_tile(2, c20, c21);
    out_tile(3, c30, c31);
}

// which epilogues get a gemm4 instantiation (each costs a 256-register kernel per operand format at compile time)
template <class Epi> struct gemm4_enabled : std::false_type {};
template <int ACT> struct gemm4_enabled<EpiStore<ACT>> : std::true_type {};
template <int G> struct gemm4_enabled<EpiResid<G>> : std::true_type {};

// (a split-K launch whose tail slice has ONE k-tile has no pipeline to run: not ok -> gemm3_launch falls back to its 128 x 128 tile)
static inline bool gemm4_ok(const Gemm3Operands& g) {
    return gemm3_ok(g) && (g.ksplit_tiles ? (g.ksplit_tiles >= 2 && (g.K / 64) % g.ksplit_tiles != 1) : g.K >= 128);
}

template <int SPLIT, class Epi>
static inline hipError_t gemm4_launch_cfg(const Gemm3Operands& g, const Epi& epi, int Z, hipStream_t st) {
    if constexpr (Epi::PAIRED || Epi::TILE || !gemm4_enabled<Epi>::value) {
        return hipErrorInvalidValue;
    } else {
        if (!gemm4_ok(g)) return hipErrorInvalidValue;
        constexpr size_t lds = 2 * (256 + 256) * 128;   // 128 KiB: one workgroup per CU
        dim3 grid(((g.N + 255) / 256) * ((g.M + 255) / 256), 1, Z);
        auto kern = gemm4_kernel<SPLIT, Epi>;
        static DevOnce once;
        hipError_t e = once.ensure([&] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, g, epi);
        return hipGetLastError();
    }
}
