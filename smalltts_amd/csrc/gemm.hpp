// LDS-tiled MFMA GEMM for gfx950:  C[M,N] (+)= A[M,K] (fp32, row-mapped) x W[N,K]^T (bf16 hi/lo)
//
// * A is fp32 in HBM (residual stream / activations stay fp32); it is split on the fly into
//   bf16 hi + bf16 lo while being staged into LDS.  W is pre-split offline (hi, lo arrays).
// * SPLIT==3: acc += Ahi*Bhi + Ahi*Blo + Alo*Bhi  (3 bf16 MFMAs, fp32 accumulate) -> ~2^-17
//   relative operand error, i.e. fp32-class results on the bf16 matrix cores.
//   SPLIT==1: plain bf16 MFMA (fast mode; error reported, not hidden).
// * v_mfma_f32_32x32x16_bf16; a wave owns TM x TN tiles of 32x32; 64-wide wavefronts, WM x WN
//   waves per workgroup.  LDS rows are padded by 16 B so ds_read_b128 fragment reads are
//   bank-conflict free (row stride 80 B / 144 B -> 16 distinct 16-B slots per 16-lane group).
// * Register-prefetched single LDS buffer: global loads of tile t+1 are issued before the MFMAs
//   of tile t.
// * Epilogues are functors fused into the accumulator write-out (bias, activation, SwiGLU,
//   gated residual, KV-cache scatter, conv-pos scatter ...).
#pragma once
#include "common.hpp"

struct GemmOperands {
    const float* A;
    RowMap amap;
    const bf16_t* Whi;
    const bf16_t* Wlo;
    long ldw;
    int M, N, K;
    long a_z, w_z;  // per-blockIdx.z element strides
    int w_zmod;     // weights use (z % w_zmod) * w_z when non-zero (grouped conv: z = batch*G + group)
    int xcd_order = 1;  // unbatched launches: XCD-aware tile order (set by gemm_launch from g_gemm_xcd; A/B switch)
};

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    bf16x4 h, l;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    l[0] = (bf16_t)(v.x - (float)h[0]);
    l[1] = (bf16_t)(v.y - (float)h[1]);
    l[2] = (bf16_t)(v.z - (float)h[2]);
    l[3] = (bf16_t)(v.w - (float)h[3]);
    hi = *reinterpret_cast<uint2*>(&h);
    lo = *reinterpret_cast<uint2*>(&l);
}


// Timeline markers (G3_STAMP*, EPI_STAMP): empty statements in the shipped library; a lab build (exp/timeline.hpp) makes them stamps.
#ifdef SMTTS_LAB
#include "exp/timeline.hpp"
#else
#define G3_STAMPK(i) do { } while (0)
#define G3_STAMP(i) do { } while (0)
#define G3_STAMPR(i) do { } while (0)
#define G3_STAMP_FENCE(v) do { } while (0)
#define EPI_STAMP(i) do { } while (0)
#define G3_TIMELINE_EXPORTS(suffix)
#endif
struct RowCtx;
// Shared accumulator write-out for gemm_kernel / gemm3_kernel.  mw0 / nw0 = first row / column of this wave.
template <int TM, int TN, class Epi>
__device__ __forceinline__ void gemm_epilogue(const Epi& epi, floatx16 (&acc)[TM][TN], int M, int N, int mw0, int nw0,
                                              int z, int lane);

template <int BM, int BN, int BK, int WM, int WN, int SPLIT, class Epi>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmOperands g, Epi epi) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int LROW = BK + 8;  // bf16 elements per LDS row (16 B pad)
    constexpr int A_CH = (BM * BK / 4) / NT;
    constexpr int W_CH = (BN * BK / 8) / NT;
    static_assert(A_CH * NT * 4 == BM * BK && W_CH * NT * 8 == BN * BK, "tile/threads mismatch");
    static_assert(!Epi::PAIRED || TN == 2, "paired epilogue needs a 32x64 wave tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sAhi = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sAlo = sAhi + BM * LROW;
    bf16_t* sWhi = sAlo + (SPLIT == 3 ? BM * LROW : 0);
    bf16_t* sWlo = sWhi + BN * LROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD has its own L2: with the plain
    // (x = column tile, y = row tile) mapping the column tiles of ONE row tile land on different XCDs and every one of them reads
    // the A rows from HBM again (counters, the codec's ConvTranspose at C = 256 / 128: 1.13 GB per launch against 0.27 GB of
    // operands + output).  Unbatched launches therefore give every XCD a contiguous run of tiles, column tile fastest.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.z == 1 && g.xcd_order) {
        const int Nt = gridDim.x, tot = Nt * (int)gridDim.y, p = bx + Nt * by;
        const int q = tot / 8, r = tot % 8, xcd = p % 8, loc = p / 8;
        const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        by = vid / Nt; bx = vid % Nt;
    }
    const int m0 = by * BM, n0 = bx * BN, z = blockIdx.z;
    const float* A = g.A + (long)z * g.a_z;
    const long wz = (long)(g.w_zmod ? z % g.w_zmod : z) * g.w_z;
    const bf16_t* Whi = g.Whi + wz;
    const bf16_t* Wlo = (SPLIT == 3) ? g.Wlo + wz : nullptr;

    // per-thread staging coordinates (fixed across k-tiles)
    long a_off[A_CH];
    int a_k[A_CH], a_lds[A_CH];
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
        int idx = c * NT + tid;
        int row = idx / (BK / 4), c4 = idx % (BK / 4);
        int m = m0 + row;
        m = m < g.M ? m : g.M - 1;
        a_off[c] = g.amap.at(m) + c4 * 4;
        a_k[c] = c4 * 4;
        a_lds[c] = row * LROW + c4 * 4;
    }
    long w_off[W_CH];
    int w_k[W_CH], w_lds[W_CH];
#pragma unroll
    for (int c = 0; c < W_CH; ++c) {
        int idx = c * NT + tid;
        int row = idx / (BK / 8), c8 = idx % (BK / 8);
        int n = n0 + row;
        n = n < g.N ? n : g.N - 1;
        w_off[c] = (long)n * g.ldw + c8 * 8;
        w_k[c] = c8 * 8;
        w_lds[c] = row * LROW + c8 * 8;
    }

    float4 ra[A_CH];
    uint4 rwh[W_CH], rwl[W_CH];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            if (k0 + a_k[c] < g.K)
                ra[c] = *reinterpret_cast<const float4*>(A + a_off[c] + k0);
            else
                ra[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < W_CH; ++c) {
            if (k0 + w_k[c] < g.K) {
                rwh[c] = *reinterpret_cast<const uint4*>(Whi + w_off[c] + k0);
                if (SPLIT == 3) rwl[c] = *reinterpret_cast<const uint4*>(Wlo + w_off[c] + k0);
            } else {
                rwh[c] = make_uint4(0, 0, 0, 0);
                if (SPLIT == 3) rwl[c] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            uint2 hi, lo;
            split4(ra[c], hi, lo);
            *reinterpret_cast<uint2*>(sAhi + a_lds[c]) = hi;
            if (SPLIT == 3) *reinterpret_cast<uint2*>(sAlo + a_lds[c]) = lo;
        }
#pragma unroll
        for (int c = 0; c < W_CH; ++c) {
            *reinterpret_cast<uint4*>(sWhi + w_lds[c]) = rwh[c];
            if (SPLIT == 3) *reinterpret_cast<uint4*>(sWlo + w_lds[c]) = rwl[c];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fk = (lane >> 5) * 8;
    const int a_frag = (wm * TM * 32 + fr) * LROW + fk;
    const int w_frag = (wn * TN * 32 + fr) * LROW + fk;

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(sAhi + a_frag + i * 32 * LROW + kk * 16);
                if (SPLIT == 3) al[i] = *reinterpret_cast<const bf16x8*>(sAlo + a_frag + i * 32 * LROW + kk * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(sWhi + w_frag + j * 32 * LROW + kk * 16);
                if (SPLIT == 3) bl[j] = *reinterpret_cast<const bf16x8*>(sWlo + w_frag + j * 32 * LROW + kk * 16);
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
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    gemm_epilogue<TM, TN, Epi>(epi, acc, g.M, g.N, m0 + wm * TM * 32, n0 + wn * TN * 32, z, lane);
}

// ------------------------------------------------------------------------------------------
// Epilogues
//
// A lane of a 32x32 MFMA tile owns ONE output column n and 16 rows m_r = mb + (r&3) + 8*(r>>2).
// Protocol: `rows()` derives everything that depends on the row only (addresses, masks, gate rows)
// once per 32-row tile; `col()` then handles one column: every global LOAD (bias, gate, residual)
// is issued before the first STORE, so the 16 elements pipeline instead of paying one memory
// round trip each (output and parameter pointers may alias as far as the compiler knows).
// ------------------------------------------------------------------------------------------
struct RowCtx {
    long off[16];
    int aux[16];
    unsigned valid;  // bit r: row in range (and not masked out, where the epilogue skips masked rows)
    int mb;          // first row of the lane's 16 (LN-fold producer: where the row partials go)
    float mu[16], rs[16];   // LN-fold consumers: mean / rstd of the lane's rows (unused, hence no registers, everywhere else)
};

// ------------------------------------------------------------------------------------------
// LN-fold (round 6): the AdaLN in front of a DiT GEMM without a norm launch (reference dit.py:19-25,197-212).
//   y = LN(x) (1 + scale) + shift,  out = y W^T + b
//     = rstd (x (1 + scale) W^T - mu W (1 + scale)) + W shift + b
// The PRODUCER of the residual row (EpiResidLN, the out-proj / FF2 epilogue) writes the operand image fp16(x (1 + scale)) and, per
// row and 32-column group, the partial sums (sum x, sum x^2); the CONSUMER (EpiQKV / EpiSwiGLU) reduces a row's NP partials in a
// fixed order when its workgroup starts (gemm3_kernel -> an LDS table of (mu, rstd) per tile row) and applies
//   rstd (acc - mu wc[n]) + wsh[n] + b[n],     wc = W (1 + scale),  wsh = W shift
// with the two per-(step, block, site) vectors from fold_vectors_kernel (kernels.hip: one launch per sampler call, t is shared by
// the batch).  Deterministic: fixed butterfly in the producer, fixed order in the consumer.
// ------------------------------------------------------------------------------------------
struct LnFoldIn {
    const float* part = nullptr;   // [M][NP][2] (sum, sum of squares) per row and 32-column group; null = fold off (plain bias epilogue)
    int NP = 0;                    // groups per row (hidden / 32)
    float inv_c = 0.f, eps = 0.f;  // 1 / hidden, LayerNorm eps
    const float* wc = nullptr;     // [N] W (1 + scale)        (null with rms)
    const float* wsh = nullptr;    // [N] W shift              (null with rms)
    int rms = 0;                   // 1: RMSNorm with weight w (the encoders, style.py:70-105 / phonemes.py:131-167): the image is x w, the consumer
                                   //    only scales by rstd = 1 / sqrt(mean x^2 + eps) — no mean, no vectors
    const float* lstat = nullptr;  // set by the kernel: LDS table [BM][2] = (mu, rstd) of the tile's rows
    int m0 = 0;                    // set by the kernel: first row of the tile
};
template <class E, class = void>
struct epi_small_n { static constexpr bool value = false; };   // epilogues of the DiT's N = 960 residual projections: extra tile shapes / ring depths are instantiated for them only
template <class E>
struct epi_small_n<E, decltype((void)E::SMALL_N)> { static constexpr bool value = E::SMALL_N; };
template <class E, class = void>
struct epi_fold_in { static constexpr bool value = false; };
template <class E>
struct epi_fold_in<E, decltype((void)E::FOLD_IN)> { static constexpr bool value = E::FOLD_IN; };
// Every value an epilogue LOADED must be complete, as far as the compiler's wait-count pass can tell, before its first STORE is
// issued: loads and stores share vmcnt on gfx9 and complete out of order with respect to each other, so a load that is still
// (or only "possibly": a skipped branch arm) pending once stores are in flight costs an `s_waitcnt vmcnt(0)` in front of EVERY
// later element — sixteen serialised store round trips, 3.4 of the 12 us a 64 x 64 tile's workgroup lived (per-k-tile timeline,
// profiles/r03p_*).  Passing the register through an empty asm makes it a plain definition: the wait lands here, once.
__device__ __forceinline__ void epi_settle(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void epi_settle(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int epi_row(int mb, int r) { return mb + (r & 3) + 8 * (r >> 2); }
// The sixteen row-mask bytes of a lane's rows.  ONE uniform branch on the pointer, then sixteen unconditional loads of clamped rows:
// written as `rowmask ? rowmask[m] : 1` per row, hipcc emitted sixteen branch diamonds with an `s_waitcnt vmcnt(0)` at every join —
// sixteen dependent round trips in front of the residual loads of every masked N = 960 projection (round 6, found in the ISA).
__device__ __forceinline__ void epi_row_masks(const uint8_t* __restrict__ rowmask, int mb, int M, int (&mk)[16]) {
    if (rowmask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            mk[r] = (int)rowmask[m < M ? m : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) epi_settle(mk[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = 1;
    }
}

// out[omap(m) + n] = mask(m) * act((acc + bias[n]) * scale)
template <int ACT>
struct EpiStore {
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = true;
    static constexpr bool TILE = false;
    float* out;
    RowMap omap;
    long o_z;
    const float* bias;     // may be null; indexed [n] (+ z*bias_z)
    long bias_z;
    float scale;
    const uint8_t* rowmask;  // may be null; [m]: masked rows are written as 0
    bf16_t* ohi;             // when non-null the result is written as a split bf16 pair (ohi/olo) instead of `out`
    bf16_t* olo;
    __device__ __forceinline__ void rows(int z, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            const bool ok = m < M;
            rc.off[r] = (long)z * o_z + omap.at(ok ? m : 0);
            rc.valid |= (ok ? 1u : 0u) << r;
        }
        epi_row_masks(rowmask, mb, M, rc.aux);   // (rows past M read row 0's byte: never stored)
    }
    __device__ __forceinline__ void col(int z, int n, const RowCtx& rc, const floatx16& acc) const {
        float b = bias ? bias[(long)z * bias_z + n] : 0.f;
        epi_settle(b);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = apply_act<ACT>((acc[r] + b) * scale);
            v[r] = rc.aux[r] ? v[r] : 0.f;
        }
        if (ohi) {   // (one decision per column, not per element)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rc.valid >> r & 1) store_act1(ohi, olo, rc.off[r] + n, v[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rc.valid >> r & 1) out[rc.off[r] + n] = v[r];
        }
    }
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}
    // ---- LDS-staged 16-bit output (gemm3 only, see gemm_epilogue_staged16) ----
    __device__ __forceinline__ bool stage16_ok() const {   // one 16-bit array, 16-byte aligned rows
        return ohi != nullptr && !sm_is_split(olo) && (omap.ld % 8) == 0 && (omap.off % 8) == 0 && (omap.bstride % 8) == 0 && (o_z % 8) == 0;
    }
    __device__ __forceinline__ bf16_t* out16() const { return ohi; }
    __device__ __forceinline__ long row_off(int z, int m) const { return (long)z * o_z + omap.at(m); }
    __device__ __forceinline__ void col16(int z, int n, const RowCtx& rc, const floatx16& acc, unsigned short (&o)[16]) const {
        const float b = bias ? bias[(long)z * bias_z + n] : 0.f;
        const bool f16 = sm_is_f16(olo);
        unsigned sat = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float v0, v1;
            if (ACT == ACT_GELU) {   // rounded to 16 bits right below: GeluQ5's 2e-6 absolute error is far inside that rounding
                v0 = gelu_q5_f((acc[r] + b) * scale); v1 = gelu_q5_f((acc[r + 1] + b) * scale);
            } else {
                v0 = apply_act<ACT>((acc[r] + b) * scale); v1 = apply_act<ACT>((acc[r + 1] + b) * scale);
            }
            v0 = rc.aux[r] ? v0 : 0.f;
            v1 = rc.aux[r + 1] ? v1 : 0.f;
            unsigned p;
            if (f16) {
                p = cvt_pk_f16_sat(v0, v1, sat);
            } else {
                bf16x4 t;
                t[0] = (bf16_t)v0; t[1] = (bf16_t)v1; t[2] = t[0]; t[3] = t[1];
                p = (*reinterpret_cast<uint2*>(&t)).x;
            }
            o[r] = (unsigned short)(p & 0xffffu);
            o[r + 1] = (unsigned short)(p >> 16);
        }
        sat_note(sat, olo);   // (rows past M / masked rows are zero or finite garbage of zero-padded operands: they cannot clamp)
    }
    __device__ __forceinline__ void colpair16(int, int, const RowCtx&, const floatx16&, const floatx16&, unsigned short (&)[16]) const {}
};

// SwiGLU on interleaved [w1 | w3] 32-column groups: out[m][nh] = silu(a + b1[nh]) * (b + b3[nh])
template <bool FOLD>   // FOLD: the LN-fold consumer code is compiled in (a separate instantiation: the plain kernels keep their registers)
struct EpiSwiGLUT {
    static constexpr bool PAIRED = true;
    static constexpr bool STAGE16 = true;
    static constexpr bool TILE = false;
    float* out;
    long ldo;
    const float* b1;  // may be null
    const float* b3;
    bf16_t* ohi;      // optional split output (see EpiStore)
    bf16_t* olo;
    static constexpr bool FOLD_IN = FOLD;
    LnFoldIn fold;    // (gemm3, FOLD only) the operand is x (1 + scale), not LN(x) (1 + scale) + shift: see LnFoldIn
    __device__ __forceinline__ void rows(int, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            rc.off[r] = (long)m * ldo;
            rc.valid |= (m < M ? 1u : 0u) << r;
        }
        if (FOLD && fold.part) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = epi_row(mb, r) - fold.m0;
                rc.mu[r] = fold.lstat[2 * lr];
                rc.rs[r] = fold.lstat[2 * lr + 1];
            }
        }
    }
    // packed [w1 | w3] column indices of hidden unit nh (32-column groups interleaved: Engine swiglu_perm)
    __device__ __forceinline__ void fold_cols(int nh, float& c1, float& c3, float& v1, float& v3) const {
        const int p1 = (nh >> 5) * 64 + (nh & 31);
        if (fold.wc) {   // (uniform; null for the RMSNorm fold)
            c1 = fold.wc[p1]; c3 = fold.wc[p1 + 32];
            v1 += fold.wsh[p1]; v3 += fold.wsh[p1 + 32];
        }
    }
    __device__ __forceinline__ void col(int, int, const RowCtx&, const floatx16&) const {}
    __device__ __forceinline__ void colpair(int, int nh, const RowCtx& rc, const floatx16& a, const floatx16& b) const {
        float v1 = b1 ? b1[nh] : 0.f, v3 = b1 ? b3[nh] : 0.f;
        float c1 = 0.f, c3 = 0.f;
        const bool fd = FOLD && fold.part != nullptr;
        if (fd) fold_cols(nh, c1, c3, v1, v3);
        epi_settle(v1); epi_settle(v3);
        if (FOLD) { epi_settle(c1); epi_settle(c3); }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rc.valid >> r & 1) {
                const float x = (fd ? rc.rs[r] * fmaf(-rc.mu[r], c1, a[r]) : a[r]) + v1;
                const float v = silu_f(x) * ((fd ? rc.rs[r] * fmaf(-rc.mu[r], c3, b[r]) : b[r]) + v3);
                if (ohi) {
                    store_act1(ohi, olo, rc.off[r] + nh, v);
                } else {
                    out[rc.off[r] + nh] = v;
                }
            }
        }
    }
    // ---- LDS-staged 16-bit output ----
    __device__ __forceinline__ bool stage16_ok() const { return ohi != nullptr && !sm_is_split(olo) && (ldo % 8) == 0; }
    __device__ __forceinline__ bf16_t* out16() const { return ohi; }
    __device__ __forceinline__ long row_off(int, int m) const { return (long)m * ldo; }
    __device__ __forceinline__ void col16(int, int, const RowCtx&, const floatx16&, unsigned short (&)[16]) const {}
    __device__ __forceinline__ void colpair16(int, int nh, const RowCtx& rc, const floatx16& a, const floatx16& b, unsigned short (&o)[16]) const {
        float v1 = b1 ? b1[nh] : 0.f, v3 = b1 ? b3[nh] : 0.f;
        float c1 = 0.f, c3 = 0.f;
        const bool fd = FOLD && fold.part != nullptr;
        if (fd) fold_cols(nh, c1, c3, v1, v3);
        const bool f16 = sm_is_f16(olo);
        unsigned sat = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float a0 = a[r], a1 = a[r + 1], g0 = b[r], g1 = b[r + 1];
            if (fd) {
                a0 = rc.rs[r] * fmaf(-rc.mu[r], c1, a0); a1 = rc.rs[r + 1] * fmaf(-rc.mu[r + 1], c1, a1);
                g0 = rc.rs[r] * fmaf(-rc.mu[r], c3, g0); g1 = rc.rs[r + 1] * fmaf(-rc.mu[r + 1], c3, g1);
            }
            const float x0 = a0 + v1, x1 = a1 + v1;
            const float w0 = silu_f(x0) * (g0 + v3), w1 = silu_f(x1) * (g1 + v3);
            unsigned p;
            if (f16) {
                p = cvt_pk_f16_sat(w0, w1, sat);
            } else {
                bf16x4 t;
                t[0] = (bf16_t)w0; t[1] = (bf16_t)w1; t[2] = t[0]; t[3] = t[1];
                p = (*reinterpret_cast<uint2*>(&t)).x;
            }
            o[r] = (unsigned short)(p & 0xffffu);
            o[r + 1] = (unsigned short)(p >> 16);
        }
        sat_note(sat, olo);
    }
};

using EpiSwiGLU = EpiSwiGLUT<false>;
using EpiSwiGLUFold = EpiSwiGLUT<true>;

// x[xmap(m) + n] += mask(m) * g(batch(m), n) * (acc + bias[n])
//   GATE 0: g = 1      GATE 1: g = gate[(grow0 + batch*grstride) * gld + n]  (table holds tanh(gate) already,
//   launch_tanh_gates)     GATE 2: g = gate[n]
template <int GATE>
struct EpiResid {
    static constexpr bool SMALL_N = GATE == 1;
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = false;
    static constexpr bool TILE = false;
    float* x;
    RowMap xmap;
    const float* bias;       // may be null
    const float* gate;       // table (GATE 1) or vector (GATE 2)
    long gld;
    int grow0, grstride;     // gate-table row of batch b = grow0 + b*grstride (GATE 1)
    int rows_per_batch;      // batch(m) = m / rows_per_batch (GATE 1)
    const uint8_t* rowmask;  // may be null: masked rows are left untouched
    __device__ __forceinline__ void rows(int, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
        int mk[16];   // all sixteen mask bytes are requested before the first is looked at (one round trip, not sixteen dependent ones)
        epi_row_masks(rowmask, mb, M, mk);
        EPI_STAMP(154);   // mask bytes landed
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            const bool ok = m < M && mk[r] != 0;
            rc.off[r] = xmap.at(ok ? m : 0);
            rc.aux[r] = GATE == 1 ? grow0 + ((ok ? m : 0) / rows_per_batch) * grstride : 0;
            rc.valid |= (ok ? 1u : 0u) << r;
        }
    }
    __device__ __forceinline__ void col(int, int n, const RowCtx& rc, const floatx16& acc) const {
        float b = bias ? bias[n] : 0.f;
        float gv[16], xv[16];
        const float g2 = GATE == 2 ? gate[n] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (rows that are not stored carry row 0's offsets: the loads are unconditional — no exec-mask diamonds — and unused)
            gv[r] = GATE == 1 ? gate[(long)rc.aux[r] * gld + n] : g2;
            xv[r] = x[rc.off[r] + n];
        }
        EPI_STAMP(155);
        epi_settle(b);   // (gv / xv are consumed in order by the store loop: settling all 32 costs registers — 115 -> 130 VGPRs on the
                         // 128 x 128 tile, one workgroup per CU less — for nothing, the first store needs them anyway)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rc.valid >> r & 1) {
                x[rc.off[r] + n] = xv[r] + gv[r] * (acc[r] + b);
            }
        }
    }
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}
};

// LN-fold producer: the gated residual of the DiT's N = 960 projections (dit.py:198,201), all batch rows sharing ONE modulation row
// (the fused sampler: t is shared), followed by what the next AdaLN'd GEMM needs instead of a norm launch (LnFoldIn):
//   x[m][n] += mask(m) gate[n] (acc + bias[n]);   y[m][n] = fmt(x[m][n] (1 + nscale[n]));   part[m][n / 32] = (sum_n x, sum_n x^2)
// Masked rows keep x but still get their image row and partials.  N % 32 == 0 (every lane of a 32-column group takes part in the
// group's butterfly).
template <int W>   // one reduce-scatter step: the lane keeps W of its 2 W values and adds the partner lane's other half (static indices only)
__device__ __forceinline__ void fold_rs_step(float (&v)[32], int lane) {
    const bool up = (lane & W) != 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const float keep = up ? v[k + W] : v[k];
        const float send = up ? v[k] : v[k + W];
        v[k] = keep + __shfl_xor(send, W, 64);
    }
}
struct EpiResidLN {
    static constexpr bool SMALL_N = true;
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = false;
    static constexpr bool TILE = false;
    float* x;
    RowMap xmap;
    const float* bias;       // may be null
    const float* gate;       // [N]: tanh(gate) of this step (modulation-table row); null: 1 (the encoders' plain residual)
    const uint8_t* rowmask;  // may be null: masked rows are left untouched
    const float* nscale;     // [N]: scale of the AdaLN in front of the NEXT GEMM (rms: the next RMSNorm's weight itself)
    bf16_t* yhi;             // that GEMM's operand image [M][yld] in the format (yhi, ylo) encode (common.hpp sm_lo_for)
    bf16_t* ylo;
    long yld;
    float* part;             // [M][NP][2]
    int NP;
    int rms = 0;             // 1: the image is x nscale (RMSNorm weight), not x (1 + nscale)
    __device__ __forceinline__ void rows(int, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
        rc.mb = mb;
        int mk[16];
        epi_row_masks(rowmask, mb, M, mk);
        EPI_STAMP(154);   // mask bytes landed
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            const bool ok = m < M;
            rc.off[r] = xmap.at(ok ? m : 0);
            rc.aux[r] = ok && mk[r] != 0;     // the residual is applied
            rc.valid |= (ok ? 1u : 0u) << r;  // the row exists
        }
    }
    __device__ __forceinline__ void col(int, int n, const RowCtx& rc, const floatx16& acc) const {
        float b = bias ? bias[n] : 0.f;
        float g = gate ? gate[n] : 1.0f, s1 = rms ? nscale[n] : 1.0f + nscale[n];
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = x[rc.off[r] + n];   // (rows past M: row 0's, unused)
        epi_settle(b); epi_settle(g); epi_settle(s1);
#pragma unroll
        for (int r = 0; r < 16; ++r) epi_settle(xv[r]);
        EPI_STAMP(155);   // residual / vector loads landed
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float xn = rc.aux[r] ? fmaf(g, acc[r] + b, xv[r]) : xv[r];
            if (rc.aux[r]) x[rc.off[r] + n] = xn;
            v[2 * r] = xn;
            v[2 * r + 1] = xn * xn;
        }
        if (sm_is_f16(ylo)) {   // (one format decision per column; the clamp count goes out once, behind the stores)
            unsigned sat = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rc.valid >> r & 1)
                    reinterpret_cast<unsigned short*>(yhi)[(long)epi_row(rc.mb, r) * yld + n] = (unsigned short)(cvt_pk_f16_sat(v[2 * r] * s1, 0.f, sat) & 0xffffu);
            sat_note(sat, ylo);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rc.valid >> r & 1) store_act1(yhi, ylo, (long)epi_row(rc.mb, r) * yld + n, v[2 * r] * s1);
        }
        EPI_STAMP(156);   // residual + image stores issued
        // reduce-scatter over the 32 lanes (columns) of this half-wave: after the step of width w a lane holds w values; lane l ends
        // with the total of value index l = 2 r + {0: sum, 1: sum of squares}
        const int lane = threadIdx.x & 63;
        fold_rs_step<16>(v, lane); fold_rs_step<8>(v, lane); fold_rs_step<4>(v, lane); fold_rs_step<2>(v, lane); fold_rs_step<1>(v, lane);
        const int l = lane & 31, r = l >> 1;
        if (rc.valid >> r & 1) part[((long)epi_row(rc.mb, r) * NP + (n >> 5)) * 2 + (l & 1)] = v[0];
    }
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}
};

// Cross-KV scatter (reference dit.py:80-93): rows m = (b, j), columns n = ((layer*2 + kv)*H + h)*dh + d
// -> dst_kv[layer][b][h][j][d]  (+bias).  K is RMS-normalised afterwards by headnorm_kernel.
struct EpiKV {
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = false;
    static constexpr bool TILE = false;
    float* kdst;
    float* vdst;
    const float* bias;  // [n]
    int B, H, dh, S;    // S = keys per batch row (R or P)
    __device__ __forceinline__ void rows(int, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            const bool ok = m < M;
            const int b = (ok ? m : 0) / S, j = (ok ? m : 0) % S;
            rc.off[r] = ((long)b * H * S + j) * dh;
            rc.valid |= (ok ? 1u : 0u) << r;
        }
    }
    __device__ __forceinline__ void col(int, int n, const RowCtx& rc, const floatx16& acc) const {
        float bv = bias[n];
        epi_settle(bv);
        int d = n % dh, t = n / dh;
        const int h = t % H;
        t /= H;
        const int kv = t & 1, layer = t >> 1;
        float* dst = kv ? vdst : kdst;
        const long cbase = ((long)layer * B * H + h) * S * dh + d;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rc.valid >> r & 1) dst[cbase + rc.off[r]] = acc[r] + bv;
    }
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}
};

// Grouped conv position embedding (reference dit.py:215-236) as per-(batch, group) GEMMs.
// z = b*G + g, m = t, n = oc (< cpg).  v = mish(acc + bias) * mask[b][t].
// FINAL == 0: write the group-major padded image  gm[z][pad + t][n]   (input of conv2)
// FINAL == 1: x[b][t][g*cpg + n] = v + h[b][t][g*cpg + n]              (dit.py:252)
template <int FINAL>
struct EpiConvPos {
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = false;
    static constexpr bool TILE = false;
    float* out;
    const float* h;       // FINAL only
    const float* bias;    // [G*cpg]
    const uint8_t* mask;  // [B][T]
    int G, cpg, T, pad, gstride;  // gstride = padded channels per group in gm image
    bf16_t* ohi;          // FINAL == 0: optional split output image
    bf16_t* olo;
    int by_group = 0;     // 0: z = b * G + g, rows = frames of one utterance; 1: z = g, rows = (b, frame) of the whole batch
    __device__ __forceinline__ void rows(int z, int mb, int M, RowCtx& rc) const {
        rc.valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = epi_row(mb, r);
            const bool ok = m < M;
            const int mm = ok ? m : 0;
            const int b = by_group ? mm / T : z / G, t = by_group ? mm - b * T : mm;
            const long zz = by_group ? (long)b * G + z : z;
            rc.aux[r] = mask[b * T + t];
            rc.off[r] = FINAL ? ((long)b * T + t) * (G * cpg) : (zz * (T + 2 * pad) + pad + t) * gstride;
            rc.valid |= (ok ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) epi_settle(rc.aux[r]);
    }
    __device__ __forceinline__ void col(int z, int n, const RowCtx& rc, const floatx16& acc) const {
        const int ch = (by_group ? z : z % G) * cpg + n;
        float bv = bias[ch];
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = (FINAL && (rc.valid >> r & 1)) ? h[rc.off[r] + ch] : 0.f;
        epi_settle(bv);
        if (FINAL)
#pragma unroll
            for (int r = 0; r < 16; ++r) epi_settle(hv[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rc.valid >> r & 1) {
                const float v = rc.aux[r] ? mish_f(acc[r] + bv) : 0.f;
                if (FINAL) {
                    out[rc.off[r] + ch] = v + hv[r];
                } else if (ohi) {
                    store_act1(ohi, olo, rc.off[r] + n, v);
                } else {
                    out[rc.off[r] + n] = v;
                }
            }
        }
    }
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}
};

// ------------------------------------------------------------------------------------------
// QKVG projection -> attention operand images (gemm3 only; attention_img.hip describes the consumer).
// Columns of the packed weight: n = (part * H + h) * HW + d, part = 0 q | 1 k | 2 v | 3 gate, every head padded from dh to HW
// (64 / 128) columns with zero weight rows, so that a 128-column tile holds whole heads of ONE part.  The whole workgroup
// passes its accumulators (+ bias) through an fp32 LDS tile and then works row-wise:
//   q / k : RMSNorm over the head (sum of squares across the wave), * norm weight, RoPE on (2i, 2i + 1), q * 1/sqrt(dh)
//           -> Q / K images [B][H][Nseq][HW]                              (dit.py:95-108, style.py:21-25,52-55)
//   v     : transposed -> V^T image [B][H][HW][Np] (lane = row: 64 consecutive keys per store instruction)
//   gate  : sigmoid -> [M][H * dh]
// all in the 16-bit operand format `prec`.  Same arithmetic, in the same order, as qkv_pack_kernel (attention_img.hip), which
// tests/test_kernels_gpu.py holds it to bit for bit.
// ------------------------------------------------------------------------------------------
template <bool FOLD>
struct EpiQKVT {
    static constexpr bool PAIRED = false;
    static constexpr bool STAGE16 = false;
    static constexpr bool TILE = true;
    static constexpr int TP = 129;   // fp32 LDS tile pitch (floats): odd, so that lane = row reads of the V part spread over the banks
    const float* bias;               // [4 * H * HW], padded like the weight rows (null: none)
    const float *qw, *kw;            // [H][dh]
    const float *rope_cos, *rope_sin;
    float eps, q_scale;
    int rot_dim, prec;
    bf16_t *q, *q_lo, *k, *k_lo, *vt, *vt_lo, *g, *g_lo;
    int Nseq, H, dh, HW, Np;
    static constexpr bool FOLD_IN = FOLD;
    LnFoldIn fold;                   // (FOLD only) the operand is x (1 + scale), not the AdaLN output: see LnFoldIn
    // (the generic column protocol is not used by this epilogue)
    __device__ __forceinline__ void rows(int, int, int, RowCtx&) const {}
    __device__ __forceinline__ void col(int, int, const RowCtx&, const floatx16&) const {}
    __device__ __forceinline__ void colpair(int, int, const RowCtx&, const floatx16&, const floatx16&) const {}

    // BN == 128; the workgroup's NW waves are laid out WM x WN with TM x TN 32x32 tiles each; `tile` = >= 64 * TP * 4 bytes of LDS
    template <int BM, int TM, int TN, int WN, int NW>
    __device__ __forceinline__ void tile_epilogue(floatx16 (&acc)[TM][TN], int M, int m0, int n0, int wave, int lane, float* tile) const {
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        const int part = n0 / (H * HW), h0 = (n0 % (H * HW)) / HW;   // (uniform)
#pragma unroll
        for (int rh = 0; rh < BM / 64; ++rh) {     // 64 rows of the tile at a time
            __syncthreads();                       // the ring (first pass) / the previous half (second pass) is no longer read
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r32 = (wm * TM + i) * 32 - rh * 64;   // first row of this 32-row block inside the half
                if (r32 < 0 || r32 >= 64) continue;              // (uniform)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = (wn * TN + j) * 32 + fr;
                    float bv = bias ? bias[n0 + c] : 0.f;
                    if (FOLD && fold.part) {   // (uniform) LN-fold: rstd (acc - mu wc) + W shift + bias
                        float cw = 0.f;
                        if (fold.wc) { cw = fold.wc[n0 + c]; bv += fold.wsh[n0 + c]; }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = r32 + 4 * fh + (r & 3) + 8 * (r >> 2);
                            const float2 st = *reinterpret_cast<const float2*>(fold.lstat + 2 * (rh * 64 + row));
                            tile[row * TP + c] = st.y * fmaf(-st.x, cw, acc[i][j][r]) + bv;
                        }
                        continue;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tile[(r32 + 4 * fh + (r & 3) + 8 * (r >> 2)) * TP + c] = acc[i][j][r] + bv;
                }
            }
            __syncthreads();
            const int mh = m0 + rh * 64;           // first row of this half
            if (part == 2) {
                // V^T: lane = row (key), the wave walks its share of the columns (dims)
                const int m = mh + lane;
                if (m < M) {
                    const int b = m / Nseq, n = m - b * Nseq;
#pragma unroll 4
                    for (int c = wave; c < 128; c += NW) {
                        const int h = h0 + c / HW, d = c % HW;
                        store_img1(vt, vt_lo, prec, (((long)b * H + h) * HW + d) * Np + n, tile[lane * TP + c]);
                    }
                }
            } else {
                // the wave's 64 / NW rows as independent chains: every load (LDS tile, rope table) is issued before the first
                // reduction, the reductions interleave, then the stores — not one global round trip per row
                constexpr int RPW = 64 / NW;
                const int c0 = 2 * lane, h = h0 + c0 / HW, d = c0 % HW;
                float x0[RPW], x1[RPW], cs[RPW], sn[RPW];
                long mrow[RPW], nrow[RPW];
                const QkPrep pp{part ? kw : qw, rope_cos, rope_sin, rot_dim, dh, eps, part ? 1.0f : q_scale};
                float w0 = 0.f, w1 = 0.f;
                if (part < 2) pp.weights(h, d, w0, w1);
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int r = wave + i * NW;
                    int m = mh + r;
                    m = m < M ? m : M - 1;         // rows past the end: computed on a valid row, not stored
                    const int b = m / Nseq, n = m - b * Nseq;
                    mrow[i] = m;
                    nrow[i] = ((long)b * H + h) * Nseq + n;
                    x0[i] = tile[r * TP + c0]; x1[i] = tile[r * TP + c0 + 1];
                    cs[i] = 1.f; sn[i] = 0.f;
                    if (part < 2) pp.rope(d, n, cs[i], sn[i]);
                }
                if (part < 2) {   // (conditional loads: complete before the first store, see epi_settle)
                    epi_settle(w0); epi_settle(w1);
#pragma unroll
                    for (int i = 0; i < RPW; ++i) { epi_settle(cs[i]); epi_settle(sn[i]); }
                }
                if (part == 3) {
#pragma unroll
                    for (int i = 0; i < RPW; ++i)
                        if (mh + wave + i * NW < M && d < dh)
                            store_img2(g, g_lo, prec, mrow[i] * ((long)H * dh) + h * dh + d, sigmoid_f(x0[i]), sigmoid_f(x1[i]));
                } else {
                    float ss[RPW];
#pragma unroll
                    for (int i = 0; i < RPW; ++i) ss[i] = fmaf(x1[i], x1[i], x0[i] * x0[i]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {   // (HW == 64: two heads per wave, the sum stays inside 32 lanes)
                        if (HW == 64 && o == 32) continue;
#pragma unroll
                        for (int i = 0; i < RPW; ++i) ss[i] += __shfl_xor(ss[i], o, 64);
                    }
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
                        pp.math(x0[i], x1[i], ss[i], w0, w1, cs[i], sn[i], d);
                        if (mh + wave + i * NW < M) {
                            if (part) store_img2(k, k_lo, prec, nrow[i] * HW + d, x0[i], x1[i]);
                            else store_img2(q, q_lo, prec, nrow[i] * HW + d, x0[i], x1[i]);
                        }
                    }
                }
            }
        }
    }
};

using EpiQKV = EpiQKVT<false>;
using EpiQKVFold = EpiQKVT<true>;

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <int TM, int TN, class Epi>
__device__ __forceinline__ void gemm_epilogue(const Epi& epi, floatx16 (&acc)[TM][TN], int M, int N, int mw0, int nw0,
                                              int z, int lane) {
    const int cn = lane & 31, rm = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = mw0 + i * 32 + rm;
        if (mb - rm >= M) continue;
        RowCtx rc;
        epi.rows(z, mb, M, rc);
        if (Epi::PAIRED) {
            const int nh = nw0 / 2 + cn;  // packed column base nw0 (64 packed = 32 outputs)
            if (nw0 + cn < N) epi.colpair(z, nh, rc, acc[i][0], acc[i][TN - 1]);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nw0 + j * 32 + cn;
                if (n < N) epi.col(z, n, rc, acc[i][j]);
            }
        }
    }
}

// 16-bit outputs (the fp16 / bf16 operand of the NEXT GEMM: GELU hidden, SwiGLU hidden) through a wave-private LDS tile.
// In the MFMA C layout a lane owns ONE column and 16 rows, so writing a [rows][cols] 16-bit array straight from the accumulators
// takes 16 two-byte stores per 32x32 tile and lane, each touching 64 B of a row: the store issue, not bandwidth, bounded the
// epilogue of the K = 512 .. 1024 products (profiles/r02*: FF1 of a codec stage 1.9x the time of its FF2 at equal flops and
// bytes).  Staged: every lane drops its converted values into the wave's [32][W] tile in LDS (W = the wave tile's output
// width), then the wave reads the tile back row-major in 16-byte pieces and stores those — W / 8 stores per row, 128-byte rows
// for a 64-wide wave tile.  `stage` = this wave's private region (32 rows x (2 W + 16) bytes; LDS ring of the finished k-loop).
template <int TM, int TN, class Epi>
__device__ __forceinline__ void gemm_epilogue_staged16(const Epi& epi, floatx16 (&acc)[TM][TN], int M, int N, int mw0, int nw0,
                                                       int z, int lane, char* stage) {
    constexpr int W = Epi::PAIRED ? 32 : TN * 32;   // output columns of one wave tile
    constexpr int RS = 2 * W + 16;                  // staged row stride in bytes (16-byte pad staggers the banks)
    constexpr int CH = W / 8;                       // 16-byte pieces per row
    const int cn = lane & 31, rm = 4 * (lane >> 5);
    const int ncol0 = Epi::PAIRED ? nw0 / 2 : nw0;  // first output column of the wave tile
    const int Nout = Epi::PAIRED ? N / 2 : N;
    bf16_t* const out = epi.out16();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mt = mw0 + i * 32;                // first row of this 32-row tile
        if (mt >= M) continue;
        RowCtx rc;
        epi.rows(z, mt + rm, M, rc);
        if (Epi::PAIRED) {
            unsigned short o[16];
            epi.colpair16(z, ncol0 + cn, rc, acc[i][0], acc[i][TN - 1], o);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<unsigned short*>(stage + (rm + (r & 3) + 8 * (r >> 2)) * RS + 2 * cn) = o[r];
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned short o[16];
                const int n = nw0 + j * 32 + cn;
                epi.col16(z, n < N ? n : N - 1, rc, acc[i][j], o);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<unsigned short*>(stage + (rm + (r & 3) + 8 * (r >> 2)) * RS + 2 * (j * 32 + cn)) = o[r];
            }
        }
        // same wave wrote and reads: LDS operations of one wave complete in order, the wait makes the data visible to the loads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < (32 * CH + 63) / 64; ++c) {
            const int id = c * 64 + lane;
            const int row = id / CH, ch = id % CH;
            const int m = mt + row, n0 = ncol0 + ch * 8;
            if (id < 32 * CH && m < M && n0 < Nout) {   // Nout % 8 == 0 is part of stage16_ok's contract (checked by the caller)
                const uint4 v = *reinterpret_cast<const uint4*>(stage + row * RS + ch * 16);
                *reinterpret_cast<uint4*>(out + epi.row_off(z, m) + n0) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is reused by the next row tile
    }
}

// ------------------------------------------------------------------------------------------
// Launcher
// ------------------------------------------------------------------------------------------
enum GemmCfg {
    CFG_64x128 = 0,   // general (wave tile 32x64) — also the SwiGLU config
    CFG_64x64 = 1,    // small N / under-filled grids
    CFG_128x128 = 2,  // large M and N
    CFG_128x32 = 3,   // tall-skinny, N <= 32
    CFG_128x64 = 4,   // tall-skinny, N <= 64
};

template <int BM, int BN, int BK, int WM, int WN, int SPLIT, class Epi>
static inline hipError_t gemm_launch_cfg(const GemmOperands& g, const Epi& epi, int Z, hipStream_t st) {
    constexpr int LROW = BK + 8;
    size_t lds = (size_t)(BM + BN) * LROW * 2 * (SPLIT == 3 ? 2 : 1);
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, Z);
    auto kern = gemm_kernel<BM, BN, BK, WM, WN, SPLIT, Epi>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, g, epi);
    return hipGetLastError();
}

static inline int gemm_pick_cfg(int M, int N, int K, bool paired) {
    if (paired) return CFG_64x128;
    if (N <= 32) return CFG_128x32;
    if (N <= 64) return M >= 4096 ? CFG_128x64 : CFG_64x64;
    long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 >= 1024) return CFG_128x128;
    long tilesL = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (tilesL >= 256 || N % 128 == 0) return CFG_64x128;
    return CFG_64x64;
}

template <int SPLIT, class Epi>
static inline hipError_t gemm_launch_split(const GemmOperands& g, const Epi& epi, int Z, int cfg, hipStream_t st) {
    const bool k32 = g.K <= 32;
    switch (cfg) {
        case CFG_64x128:
            return gemm_launch_cfg<64, 128, 64, 2, 2, SPLIT, Epi>(g, epi, Z, st);
        case CFG_128x128:
            if constexpr (!Epi::PAIRED) return gemm_launch_cfg<128, 128, 64, 2, 2, SPLIT, Epi>(g, epi, Z, st);
            break;
        case CFG_64x64:
            if constexpr (!Epi::PAIRED) return gemm_launch_cfg<64, 64, 64, 2, 2, SPLIT, Epi>(g, epi, Z, st);
            break;
        case CFG_128x32:
            if constexpr (!Epi::PAIRED) return gemm_launch_cfg<128, 32, 64, 4, 1, SPLIT, Epi>(g, epi, Z, st);
            break;
        case CFG_128x64:
            if constexpr (!Epi::PAIRED) {
                if (k32) return gemm_launch_cfg<128, 64, 32, 4, 1, SPLIT, Epi>(g, epi, Z, st);
                return gemm_launch_cfg<128, 64, 64, 4, 1, SPLIT, Epi>(g, epi, Z, st);
            }
            break;
    }
    return hipErrorInvalidValue;
}

// split: 1 = plain bf16, 3 = split-bf16 (fp32-class).  This fp32-A kernel serves the cold sites only and has no fp16
// variant: PREC_F16 runs as split-bf16 here (its OUTPUT may still be written in any operand format, see store_act1).
template <class Epi>
static inline hipError_t gemm_launch(const GemmOperands& g_in, const Epi& epi, int Z, int split, hipStream_t st,
                                     int cfg = -1) {
    if (g_in.M <= 0 || g_in.N <= 0) return hipSuccess;
    extern int g_gemm_xcd;
    GemmOperands g = g_in;
    g.xcd_order = g_gemm_xcd;
    if (cfg < 0) cfg = gemm_pick_cfg(g.M, g.N, g.K, Epi::PAIRED);
    if (split != PREC_BF16) return gemm_launch_split<3, Epi>(g, epi, Z, cfg, st);
    return gemm_launch_split<1, Epi>(g, epi, Z, cfg, st);
}
