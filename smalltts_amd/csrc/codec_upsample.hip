// Streaming ConvTranspose1d(k = 2r, stride r) for the two finest codec stages (C_in = 128 -> 64 and 64 -> 32, r = 2), gfx950:
//     out[t * r + j][co] = bias[j * C_out + co] + sum_k W[j * C_out + co][k] * A[t][k],   A[t] = (x[t-1] | x[t])  (K = 2 C_in)
// i.e. the GEMM  Out[M][N = r C_out] = A[M][K] . W^T  over overlapping rows of the padded channels-last image
// (engine.hip, codec_decode).  At these stages the product is pure streaming (K = 256 / 128, 0.5 GB in + out per launch),
// and the general GEMM kernels ran it at 2 TB/s.  Here the whole W sits in LDS as split bf16 (128 / 32 KiB) for the life of a
// persistent workgroup and every wave runs on its own, 32 rows at a time (codec_ffn_wave.hip's first product):
//   Out^T[n][row] = W . A^T     A operand = W fragments (LDS), B operand = the wave's rows, read straight from global x
//                               (lane = row, 8 consecutive k per k16 half), split to bf16 hi / lo in registers.
// No workgroup barrier, no LDS traffic for activations; the raw rows of the NEXT k16 steps are in flight while the current
// step's MFMAs run.  Every lane ends up with its own output row: 4 consecutive n per accumulator group -> 16-B stores.
#include "gemm3.hpp"
#include "kernels.hpp"
#include "prof.hpp"

extern int g_persist_mask;   // engine.hip: which persistent kernels the throughput-mode grid cap applies to (1 streamed FFN, 2 one-pass / wave FFN, 4 upsample)

namespace {
typedef float f32x2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16u(float a, float b) {
    f32x2u v;
    v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2u));
}
__device__ __forceinline__ void split_pair_u(float a, float b, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16u(a, b);
    lo = cvt_pk_bf16u(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

struct UpsampleArgs {
    const float* x;
    RowMap amap;           // row m of A (K contiguous floats: frames t-1, t)
    const bf16_t* whi;     // [N][ldw]
    const bf16_t* wlo;
    int ldw;
    const float* bias;     // [N] or null
    float* out;
    RowMap omap;           // row m of Out (N contiguous floats: r output frames)
    int M;
};

template <int K, int N, int SPLIT>
__global__ __launch_bounds__(512) void codec_upsample_wave_kernel(UpsampleArgs a) {
    constexpr int KK = K / 16;             // k16 steps
    constexpr int NT = N / 32;             // 32-row output (n) tiles
    constexpr int RB = 2 * K;              // bytes per W row image (256 or 512)
    constexpr int CPR = RB / 16;           // 16-B chunks per row (16 or 32)
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
    constexpr int W_ARR = N * RB;
    constexpr int OFF_B = NARR * W_ARR;    // bias[N] (fp32)
    // raw rows are loaded PF k16 steps ahead, across tile boundaries (the ring never drains).  K = 256: a whole tile ahead (16 steps:
    // 179 -> 164 us against 8, profiles/r03ap_*); K = 128 with its W fragments hoisted into 128 registers spills with a deeper ring.
    // K = 128: a whole tile ahead too (8 steps), with the W fragments re-read from LDS per tile instead of living in 128 registers
    // (-DUP_OPAQUE128=0: the round-2 form, 4 steps ahead and hoisted fragments): 151.5 -> 143.5 us (profiles/r03ar_*)
#ifndef UP_OPAQUE128
#define UP_OPAQUE128 1
#endif
    constexpr int PF = KK >= 16 ? 16 : (UP_OPAQUE128 ? 8 : 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // ---- one-time: W image into LDS; 16-B chunk c of row r stored at (c & ~15) | ((c ^ r) & 15): conflict-free fragment reads ----
    for (int i = tid; i < N * CPR; i += 512) {
        const int r = i / CPR, c = i % CPR;
        const int dst = r * RB + (((c & ~15) | ((c ^ r) & 15)) << 4);
        *reinterpret_cast<uint4*>(smem + dst) = *reinterpret_cast<const uint4*>(a.whi + (long)r * a.ldw + c * 8);
        if (SPLIT == 3) *reinterpret_cast<uint4*>(smem + W_ARR + dst) = *reinterpret_cast<const uint4*>(a.wlo + (long)r * a.ldw + c * 8);
    }
    float* vb = reinterpret_cast<float*>(smem + OFF_B);
    for (int i = tid; i < N; i += 512) vb[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();

    // W fragment of (n tile nt, k16 step kk): row 32 nt + fr, chunk c = 2 kk + fh.  (c ^ row) & 15 = (2 kk & 15) ^ ((fh ^ fr) & 15)
    // and row * RB has no bits below 256, so the offset is (w_a0 ^ ((2 kk & 15) << 4)) + (2 kk >> 4) * 256 + nt * 32 * RB
    const int w_a0 = fr * RB + (((fh ^ fr) & 15) << 4);

    const int ntiles = (a.M + 31) / 32;
    const int wg = blockIdx.x * 8 + wave, nwg = gridDim.x * 8;
    auto row_ptr = [&](int wt) {
        int m = wt * 32 + fr;
        m = m < a.M ? m : a.M - 1;
        return a.x + a.amap.at(m) + 8 * fh;  // this lane's 8 k of step kk: + 16 kk
    };
    float4 xa[PF][2];  // ring of raw k16 steps in flight (compile-time indices: kk % PF; KK % PF == 0)
    static_assert(KK % PF == 0, "ring slots must line up across tiles");
    const float* xr_next = row_ptr(wg < ntiles ? wg : 0);
#pragma unroll
    for (int kk = 0; kk < PF; ++kk) {
        xa[kk][0] = *reinterpret_cast<const float4*>(xr_next + 16 * kk);
        xa[kk][1] = *reinterpret_cast<const float4*>(xr_next + 16 * kk + 4);
    }
#pragma unroll 1
    for (int wt = wg; wt < ntiles; wt += nwg) {
        const int m_cur = wt * 32 + fr;
        // The W fragments are the same for every tile.  With K = 128, N = 64 hipcc hoists their LDS reads out of this loop and
        // keeps all 128 registers of them (no LDS traffic at all); at K = 256, N = 128 they would need 512 registers, so the
        // address is made opaque per tile and the reads stay inside the loop.
        int wa = w_a0;
        if (KK * NT * NARR * 4 > 160 || UP_OPAQUE128) asm volatile("" : "+v"(wa));
        const float* xr = xr_next;
        xr_next = row_ptr(wt + nwg < ntiles ? wt + nwg : wt);  // (last tile: harmless re-read of its own first steps)
        floatx16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)  // accumulators start from the bias: row r of tile nt is n = 32 nt + (r & 3) + 8 (r >> 2) + 4 fh
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(vb + 32 * nt + 8 * q + 4 * fh);
                acc[nt][4 * q + 0] = bv.x; acc[nt][4 * q + 1] = bv.y; acc[nt][4 * q + 2] = bv.z; acc[nt][4 * q + 3] = bv.w;
            }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float4 v0 = xa[kk % PF][0], v1 = xa[kk % PF][1];
            {   // refill the slot: step kk + PF of this tile, or step kk + PF - KK of the next one
                const float* src = kk + PF < KK ? xr + 16 * (kk + PF) : xr_next + 16 * (kk + PF - KK);
                xa[kk % PF][0] = *reinterpret_cast<const float4*>(src);
                xa[kk % PF][1] = *reinterpret_cast<const float4*>(src + 4);
            }
            unsigned hp[4], lp[4];
            split_pair_u(v0.x, v0.y, hp[0], lp[0]);
            split_pair_u(v0.z, v0.w, hp[1], lp[1]);
            split_pair_u(v1.x, v1.y, hp[2], lp[2]);
            split_pair_u(v1.z, v1.w, hp[3], lp[3]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(hp));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(lp));
            const char* wb = smem + (wa ^ (((2 * kk) & 15) << 4)) + ((2 * kk) >> 4) * 256;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wb + nt * 32 * RB);
                if (SPLIT == 3) {
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wb + nt * 32 * RB + W_ARR);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc[nt], 0, 0, 0);
                }
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc[nt], 0, 0, 0);
            }
        }
        // (tried: settling the ring's prefetched steps before these stores, so that the next tile does not start with an s_waitcnt
        // vmcnt(0) behind them — 151 -> 175 us at K = 128: waiting for the prefetch here costs more than the drain there, r03r)
        if (m_cur < a.M) {
            float* orow = a.out + a.omap.at(m_cur);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(orow + 32 * nt + 8 * q + 4 * fh) =
                        make_float4(acc[nt][4 * q + 0], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]);
        }
    }
}

template <int K, int N, int SPLIT>
hipError_t upsample_go(const UpsampleArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * N * 2 * K + (size_t)N * 4;
    static_assert(lds <= 160 * 1024, "weights must fit LDS");
    auto kern = codec_upsample_wave_kernel<K, N, SPLIT>;
    static DevOnce once;
    int cus = 256;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }, &cus);
    if (e != hipSuccess) return e;
    if (!(g_persist_mask & 4)) cus = once.real_cus();   // (A/B: which persistent kernels the throughput-mode grid cap applies to)
    const int ntiles = (a.M + 31) / 32;
    const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
    int grid = (ntiles + 7) / 8;
    grid = grid < cus * per_cu ? grid : cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}
}  // namespace

bool codec_upsample_wave_ok(int K, int N) { return (K == 256 && N == 128) || (K == 128 && N == 64); }

// A rows: amap (K contiguous floats each); W [N][ldw] split bf16; out rows: omap (N contiguous floats each)
hipError_t launch_codec_upsample_wave(const float* x, RowMap amap, const bf16_t* whi, const bf16_t* wlo, int ldw, const float* bias,
                                      float* out, RowMap omap, int M, int K, int N, int split, hipStream_t st) {
    if (!codec_upsample_wave_ok(K, N) || amap.ld % 4 || amap.off % 4 || omap.ld % 4 || omap.off % 4 || ldw % 8 ||
        (amap.rpb && amap.bstride % 4) || (omap.rpb && omap.bstride % 4))
        return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    UpsampleArgs a{x, amap, whi, wlo, ldw, bias, out, omap, M};
    ProfScope ps(st, K == 256 ? "codec_upsample_wave<256x128>" : "codec_upsample_wave<128x64>", 2.0 * M * (double)K * N,
                 4.0 * M * (K / 2 + N) + 4.0 * (double)K * N);
    if (K == 256) return split == 3 ? upsample_go<256, 128, 3>(a, st) : upsample_go<256, 128, 1>(a, st);
    return split == 3 ? upsample_go<128, 64, 3>(a, st) : upsample_go<128, 64, 1>(a, st);
}
