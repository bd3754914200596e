// Fused codec FFN for the narrow (C <= 128) stages of the decoder/encoder:
//     x += ffn_gamma * ( W2 . gelu( W1 . RMSNorm(x; g, eps) + b1 ) + b2 )
// One workgroup (4 waves, 2x2) owns 64 frames.  The 4C-wide hidden never leaves the CU: it is produced in
// 64-column chunks (GEMM1 -> bias -> GELU -> split bf16 -> LDS) and immediately consumed as the A operand of
// GEMM2, whose 64 x C accumulator stays in registers for the whole kernel.  Per block the HBM traffic is one read
// and one write of x (the unfused path moved ~13 |x| per block).  W1 / W2 chunks (L2-resident, shared by all
// workgroups) arrive by direct-to-LDS DMA with the same XOR-swizzled 128-B-row image as gemm3; W1 is double
// buffered so its load overlaps a whole chunk of MFMA work.
// CP = channel count padded to a multiple of 64 (C = 32 runs as CP = 64 with zero-padded weights).
#include "gemm3.hpp"
#include "kernels.hpp"
#include "prof.hpp"

int g_ffn_bm128 = 1;  // C = 128: 1 = 128-frame tiles / 8 waves (default), 0 = 64-frame tiles / 4 waves (A/B)

struct FfnArgs {
    float* x;
    RowMap img;            // row m of x
    const float* norm_w;   // [C]
    const bf16_t* w1hi;    // [F][CP]   (K zero-padded to CP)
    const bf16_t* w1lo;
    const float* b1;       // [F]
    const bf16_t* w2hi;    // [CP][F]   (rows >= C are zero)
    const bf16_t* w2lo;
    const float* b2;       // [C]
    const float* gamma;    // [C]
    int M, C, F;
    float eps;
};

// BM = 64: 4 waves; BM = 128: 8 waves (two per SIMD) and half the L2 -> LDS weight traffic per frame.
// W1B = 2: W1 double-buffered (W1_{j+1} loads during all of chunk j).  W1B = 1 (needed at CP = 128, BM = 128: 160 KiB
// exactly): W1_{j+1} is issued once all waves left GEMM1_j (extra barrier) and streams in under GELU_j + GEMM2_j.
template <int CP, int SPLIT, int BM, int W1B>
__global__ __launch_bounds__(BM * 4) void codec_ffn_kernel(FfnArgs a) {
    constexpr int KT = CP / 64;                 // k-tiles of GEMM1
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
    constexpr int NWV = BM / 16;                // waves: (BM / 32) x 2
    constexpr int NT = NWV * 64;
    constexpr int TILE = 64 * 128;              // bytes of one [64 rows][64 bf16] k-tile image (weights)
    constexpr int RT = BM * 128;                // bytes of one [BM rows][64 bf16] k-tile image (activations)
    constexpr int N_ARR = KT * RT;              // n tile, per array
    constexpr int W1_ARR = KT * TILE;           // W1 chunk (64 hidden rows x CP), per array
    constexpr int H_ARR = RT;                   // h chunk (BM rows x 64 hidden), per array
    constexpr int W2_ARR = CP * 128;            // W2 chunk (CP out rows x 64 k), per array
    constexpr int OFF_N = 0;
    constexpr int OFF_W1 = OFF_N + NARR * N_ARR;
    constexpr int OFF_H = OFF_W1 + W1B * NARR * W1_ARR;
    constexpr int OFF_W2 = OFF_H + NARR * H_ARR;
    constexpr int TN2 = CP / 64;                // 32-col tiles per wave in GEMM2 (wave tile 32 x CP/2)
    constexpr int W1_PW = (NARR * KT * 8) / NWV;  // DMA slots per wave per W1 chunk
    constexpr int W2_PW = (NARR * (CP / 8)) / NWV;
    static_assert(W1_PW * NWV == NARR * KT * 8 && W2_PW * NWV == NARR * (CP / 8), "DMA slots must divide over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int NJ = a.F / 64;
    const unsigned lds0 = (unsigned)(size_t)SM_LPTR(smem);

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
    // W1 chunk j: hidden rows j*64 .. +64, all CP k.  slot -> (array, k-tile, 8-row block)
    auto issue_w1 = [&](int j, int buf) {
#pragma unroll
        for (int i = 0; i < W1_PW; ++i) {
            const int slot = wave * W1_PW + i;
            const int arr = slot / (KT * 8), rem = slot % (KT * 8), kt = rem / 8, rb = rem % 8;
            const int r = rb * 8 + (lane >> 3), p = lane & 7, c = p ^ ((r >> 1) & 7);
            const bf16_t* src = (arr ? a.w1lo : a.w1hi) + (long)(j * 64 + r) * CP + kt * 64 + c * 8;
            dma16(src, lds0 + (unsigned)__builtin_amdgcn_readfirstlane(OFF_W1 + (buf * NARR + arr) * W1_ARR + kt * TILE + rb * 1024));
        }
    };
    // W2 chunk j: all CP out rows, k = hidden j*64 .. +64
    auto issue_w2 = [&](int j) {
#pragma unroll
        for (int i = 0; i < W2_PW; ++i) {
            const int slot = wave * W2_PW + i;
            const int arr = slot / (CP / 8), rb = slot % (CP / 8);
            const int r = rb * 8 + (lane >> 3), p = lane & 7, c = p ^ ((r >> 1) & 7);
            const bf16_t* src = (arr ? a.w2lo : a.w2hi) + (long)r * a.F + j * 64 + c * 8;
            dma16(src, lds0 + (unsigned)__builtin_amdgcn_readfirstlane(OFF_W2 + arr * W2_ARR + rb * 1024));
        }
    };

    issue_w1(0, 0);

    // ---- phase 0: RMSNorm of the 64-frame tile -> split bf16 A image in LDS ---------------------------
    {
        const int C4 = a.C >> 2;                 // float4 per row (8, 16 or 32)
        const int lpr = C4;                      // lanes per row
        const int rows_per_pass = NT / lpr;
        for (int r0 = 0; r0 < BM; r0 += rows_per_pass) {
            const int r = r0 + tid / lpr, c4 = tid % lpr;
            const int m = m0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < a.M) v = reinterpret_cast<const float4*>(a.x + a.img.at(m))[c4];
            float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            for (int o = lpr >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rstd = 1.0f / sqrtf(ss / (float)a.C + a.eps);
            const float4 g = reinterpret_cast<const float4*>(a.norm_w)[c4];
            const float o4[4] = {v.x * rstd * g.x, v.y * rstd * g.y, v.z * rstd * g.z, v.w * rstd * g.w};
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (bf16_t)o4[e];
                l[e] = (bf16_t)(o4[e] - (float)h[e]);
            }
            const int k = c4 * 4, kt = k >> 6, kc = (k & 63) >> 3;
            const int off = kt * RT + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4) + (k & 7) * 2;
            *reinterpret_cast<bf16x4*>(smem + OFF_N + off) = h;
            if (SPLIT == 3) *reinterpret_cast<bf16x4*>(smem + OFF_N + N_ARR + off) = l;
        }
        if (a.C < CP) {  // zero the padded k columns [C, CP) of the n tile (C = 32 -> chunks 4..7 of every row)
            for (int i = tid; i < BM * 4; i += NT) {
                const int r = i >> 2, kc = 4 + (i & 3);
                const int off = r * 128 + ((kc ^ ((r >> 1) & 7)) << 4);
                *reinterpret_cast<uint4*>(smem + OFF_N + off) = make_uint4(0, 0, 0, 0);
                if (SPLIT == 3) *reinterpret_cast<uint4*>(smem + OFF_N + N_ARR + off) = make_uint4(0, 0, 0, 0);
            }
        }
    }

    // fragment offsets
    const int fr = lane & 31, fh = lane >> 5;
    int a1_off[4], b1_off[4], a2_off[4], b2_off[TN2][4];
    {
        const int ra = wm * 32 + fr, rb = wn * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a1_off[kk] = ra * 128 + (((kk * 2 + fh) ^ ((ra >> 1) & 7)) << 4);
            b1_off[kk] = rb * 128 + (((kk * 2 + fh) ^ ((rb >> 1) & 7)) << 4);
            a2_off[kk] = a1_off[kk];
#pragma unroll
            for (int t = 0; t < TN2; ++t) {
                const int r2 = (wn * TN2 + t) * 32 + fr;
                b2_off[t][kk] = r2 * 128 + (((kk * 2 + fh) ^ ((r2 >> 1) & 7)) << 4);
            }
        }
    }

    floatx16 acc2[TN2];
#pragma unroll
    for (int t = 0; t < TN2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    for (int j = 0; j < NJ; ++j) {
        const int buf = W1B == 2 ? (j & 1) : 0;
        wait_vmcnt<0>();                 // W1_j (the only DMA of this wave still in flight) has landed
        // a raw s_barrier does not wait for this wave's own ds_writes (n tile in phase 0): drain lgkmcnt first,
        // otherwise another wave can pass the barrier and read the tile before the writes land (seen as sparse
        // run-to-run differences of ~1e-4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // B1: n tile written (j = 0); GEMM2_{j-1} done -> h and W2 buffers free
        issue_w2(j);
        if (W1B == 2 && j + 1 < NJ) issue_w1(j + 1, buf ^ 1);

        // ---- GEMM1: h_j[64 x 64] = n[64 x CP] . W1_j[64 x CP]^T  (wave tile 32 x 32) -----------------
        floatx16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        const char* nb = smem + OFF_N;
        const char* w1b = smem + OFF_W1 + buf * NARR * W1_ARR;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(nb + kt * RT + a1_off[kk]);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w1b + kt * TILE + b1_off[kk]);
                if (SPLIT == 3) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(nb + N_ARR + kt * RT + a1_off[kk]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w1b + W1_ARR + kt * TILE + b1_off[kk]);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc1, 0, 0, 0);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
            }
        if (W1B == 1) {  // single W1 buffer: refill it as soon as every wave has left GEMM1_j (reads consumed by the MFMAs)
            __builtin_amdgcn_s_barrier();
            if (j + 1 < NJ) issue_w1(j + 1, 0);
        }
        // ---- bias + GELU -> split bf16 -> h image (A operand of GEMM2): element (row, k = wn*32 + col) ----
        {
            const int k = wn * 32 + (lane & 31);
            const float bv = a.b1[j * 64 + k];
            const int kc = k >> 3, ko = (k & 7) * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = gelu_f(acc1[r] + bv);
                bf16_t hh, ll;
                split1(v, hh, ll);
                const int off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4) + ko;
                *reinterpret_cast<bf16_t*>(smem + OFF_H + off) = hh;
                if (SPLIT == 3) *reinterpret_cast<bf16_t*>(smem + OFF_H + H_ARR + off) = ll;
            }
        }
        // W2_j was issued before W1_{j+1}: it has landed once at most W1_PW younger DMAs remain in flight
        if (j + 1 < NJ)
            wait_vmcnt<W1_PW>();
        else
            wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's h-tile ds_writes have landed
        __builtin_amdgcn_s_barrier();    // B2: h_j complete, everyone's W2_j landed

        // ---- GEMM2: out[64 x CP] += h_j[64 x 64] . W2_j[CP x 64]^T  (wave tile 32 x CP/2) ------------
        const char* hb = smem + OFF_H;
        const char* w2b = smem + OFF_W2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(hb + a2_off[kk]);
            bf16x8 al;
            if (SPLIT == 3) al = *reinterpret_cast<const bf16x8*>(hb + H_ARR + a2_off[kk]);
#pragma unroll
            for (int t = 0; t < TN2; ++t) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w2b + b2_off[t][kk]);
                if (SPLIT == 3) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w2b + W2_ARR + b2_off[t][kk]);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2[t], 0, 0, 0);
                }
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2[t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: x += gamma * (out + b2)  (loads first, then stores) ------------------------------------
    {
        const int cn = lane & 31, rm = 4 * (lane >> 5);
        long off[16];
        unsigned valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + rm + (r & 3) + 8 * (r >> 2);
            const bool ok = m < a.M;
            off[r] = a.img.at(ok ? m : 0);
            valid |= (ok ? 1u : 0u) << r;
        }
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int n = (wn * TN2 + t) * 32 + cn;
            if (n >= a.C) continue;
            const float bv = a.b2[n], gv = a.gamma[n];
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = (valid >> r & 1) ? a.x[off[r] + n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (valid >> r & 1) a.x[off[r] + n] = xv[r] + gv * (acc2[t][r] + bv);
        }
    }
}

template <int CP, int SPLIT, int BM, int W1B>
static hipError_t ffn_go(const FfnArgs& a, hipStream_t st) {
    constexpr int KT = CP / 64, NARR = SPLIT == 3 ? 2 : 1, TILE = 64 * 128, RT = BM * 128;
    constexpr size_t lds = (size_t)NARR * (KT * RT + W1B * KT * TILE + RT + CP * 128);
    static_assert(lds <= 160 * 1024, "fused FFN LDS budget");
    auto kern = codec_ffn_kernel<CP, SPLIT, BM, W1B>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(BM * 4), lds, st, a);
    return hipGetLastError();
}

// C in {32, 64, 128}; weights packed for CP = max(C, 64): w1 [F][CP], w2 [CP][F]
hipError_t launch_codec_ffn_fused(float* x, RowMap img, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo,
                                  const float* b1, const bf16_t* w2hi, const bf16_t* w2lo, const float* b2,
                                  const float* gamma, int M, int C, int F, float eps, int split, hipStream_t st) {
    if (!(C == 32 || C == 64 || C == 128) || F % 64 || img.ld % 4 || img.off % 4) return hipErrorInvalidValue;
    FfnArgs a{x, img, norm_w, w1hi, w1lo, b1, w2hi, w2lo, b2, gamma, M, C, F, eps};
    const int CP = C < 64 ? 64 : C;
    // algorithmic: two GEMMs; bytes: x read + written once, weights once
    ProfScope ps(st, C == 128 ? "codec_ffn_fused<128>" : C == 64 ? "codec_ffn_fused<64>" : "codec_ffn_fused<32>",
                 4.0 * M * (double)C * F, 8.0 * M * C + 8.0 * (double)C * F);
    // CP = 64 (C = 64, 32): 64-frame tiles, two independent 4-wave workgroups per CU (80 KiB each) — measured faster
    // than one 128-frame / 8-wave workgroup (1.59 vs 1.74 ms and 1.92 vs 2.17 ms per batch)
    if (CP == 64) return split == 3 ? ffn_go<64, 3, 64, 2>(a, st) : ffn_go<64, 1, 64, 2>(a, st);
    if (g_ffn_bm128) return split == 3 ? ffn_go<128, 3, 128, 1>(a, st) : ffn_go<128, 1, 128, 1>(a, st);
    return split == 3 ? ffn_go<128, 3, 64, 2>(a, st) : ffn_go<128, 1, 64, 2>(a, st);
}
