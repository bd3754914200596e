// Wave-autonomous fused codec FFN for the narrowest stages (C = 32, 64), gfx950:
//     x += ffn_gamma * ( W2 . gelu( W1 . RMSNorm(x; g, eps) + b1 ) + b2 )
//
// At C <= 64 both weight matrices fit in LDS as split bf16 (16 C^2 bytes: 16 / 64 ... 128 KiB), so nothing is
// streamed: a persistent workgroup loads them once and then every WAVE runs on its own — 32 frames at a time, no
// workgroup barrier, no LDS traffic for activations:
//   H^T[hidden][frame] = W1 . N^T    A = W1 fragments (LDS), B = the wave's normalised frames, built in registers
//                                    straight from global x (lane = frame, 8 consecutive channels per k16 half);
//   GELU on the accumulators; because the product is transposed every lane owns ONE frame column, and the
//   accumulator rows it holds are exactly what v_permlane32_swap turns into the B fragments of the second product
//   (same trick as attention_mfma.hip), so the 4C-wide hidden never leaves the register file;
//   Out^T[channel][frame] += W2 . H^T   A = W2 fragments (LDS), one 32-row hidden tile at a time.
// The hidden tiles are software-pipelined inside each wave (see the kernel): two waves that share a SIMD do not overlap one
// wave's GELU with the other's MFMAs by themselves (measured: tools/ubench/mfma_valu.hip), and packed fp32 instructions never
// run next to MFMAs at all.
#include "gemm3.hpp"
#include "kernels.hpp"
#include "prof.hpp"
#include <type_traits>

extern int g_persist_mask;   // engine.hip: which persistent kernels the throughput-mode grid cap applies to (1 streamed FFN, 2 one-pass / wave FFN, 4 upsample)

typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pair in one v_cvt_pk_bf16_f32; split_pair also returns the packed bf16 of the two residuals
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2 v;
    v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16(a, b);
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// Sum of squares of four values on top of `acc`, spelled out (one multiply, three FMAs, one add; nothing left for the compiler to
// contract): every instantiation of the kernel body (strided one-pass block, stage chain of 1 / 2 / 3 blocks) must perform the SAME
// fp32 operations.  With `a * a + b * b + ...`, `sum / C + eps` and `x += g * (acc + b)` left to -ffp-contract=fast the chain's
// blocks differed from the one-pass block by an ulp here and there (which the fp16 operand roundings then amplify to 1e-4), and
// "one launch per stage" must not change a bit — those three places are written with fmaf now.
__device__ __forceinline__ float ssq4(const float4& v, float acc) {
    return acc + fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
}

struct FfnWaveArgs {
    float* x;
    RowMap img;            // row m of x
    const float* norm_w;   // [C]
    const bf16_t* w1hi;    // [F][ld1]  (first C columns used)
    const bf16_t* w1lo;
    int ld1;
    const float* b1;       // [F]
    const bf16_t* w2hi;    // [>= C][F]
    const bf16_t* w2lo;
    const float* b2;       // [C]
    const float* gamma;    // [C]
    int M;
    float eps;
    // fused mixer (MIX kernels only): x_mid = xin + mgamma (mnorm_w * conv7(RMSNorm_noaffine(xin)) + dw_b), then the FFN on x_mid,
    // result written to x (a DIFFERENT image: tiles read 6 halo frames their neighbours overwrite)
    const float* xin;
    const float* mnorm_w;  // [C]
    const float* dw_w;     // [7][C]
    const float* dw_b;     // [C]
    const float* mgamma;   // [C]
};

// Stage chain (round 5): the NB blocks of one codec stage back to back on every 32-frame tile, the image read once and written once
// per STAGE instead of once per block.  b[i] = block i's weights / vectors; b[0].xin / b[0].x = the stage's input / output image.
struct FfnChainArgs {
    FfnWaveArgs b[3];
    int tpu;               // 32-frame tiles per utterance (img.rpb / 32): the blocks' causal halos restart there
};

// NWV = waves per workgroup.  The waves never synchronise after the weights are in LDS, so the workgroup size only sets how many
// waves share one LDS copy of the weights and how the register file divides.  Measured (profiles/r02f_ab_wave_occupancy.txt):
// more waves per SIMD do NOT help these kernels — C = 32 with 4-wave workgroups (116 registers, 16 waves per CU) 175 us against
// 157 us with 8-wave workgroups (8 waves per CU); C = 64 forced to 168 registers (3 waves per SIMD, 208 B of scratch) 294 us
// against 184 us — so both keep one 8-wave workgroup per CU (the macros stay for A/B builds).
#ifndef FW_NWV64
#define FW_NWV64 8
#endif
#ifndef FW_MINW64
#define FW_MINW64 2
#endif
// MIX: the whole codec block in one pass over the image — the mixer (RMSNorm -> causal depthwise conv k = 7 -> LayerScale residual)
// is computed for the wave's 32 frames in front of the FFN, so a block reads the image once and writes it once instead of
// twice each (mixer_fused + this kernel): at C = 32 / 64 both were HBM-bound.  The conv runs over TIME, i.e. across lanes
// (lane = frame): every lane drops its normalised frame into a wave-private [38][C + 4] fp32 tile in LDS (rows 0..5 = the six
// halo frames in front of the tile, loaded and normalised by the first 6 C / 8 lanes), then accumulates its own channels over
// the seven rows fr .. fr + 6.  Requires a tile to lie inside one batch item (T % 32 == 0) and K = 7; the engine falls back
// to the two-kernel path otherwise.
// C = 32 at the single-array formats: with the packed-fp16 GELU the kernel is no longer VALU-bound (SQ counters, profiles/r04x_*: VALU
// instructions -20 %, waiting 27 -> 43 % of wave cycles) and its 132 registers are four short of letting a second 8-wave
// workgroup share the CU (LDS has room for two): asking for four waves per SIMD makes hipcc fit it into 128 (FW_MINW32=1 restores).
#ifndef FW_MINW32
#define FW_MINW32 4
#endif
// NB > 1 (MIX only) = the stage chain.  A wave then walks a CONTIGUOUS run of tiles of one utterance in time order and runs block
// 0 .. NB - 1 on each tile before the next: block i + 1's input tile is block i's output, still in registers (accumulator layout ->
// B-fragment layout by v_permlane32_swap, the inverse of KEEPX's rearrangement), and the six causal halo frames block i + 1's mixer
// needs in front of the tile are the last six frames of block i's output on the PREVIOUS tile, which the same wave parked in LDS
// (raw: they are normalised where they are used, by the code that normalises block 0's halo from the image — bit-identical
// arithmetic to one launch per block).  A run that does not start at an utterance start first runs the tile in front of it without
// storing: block i's output there is right from frame 6 i on, in particular on the last six frames (the next tile's halos).
template <int C, int SPLIT, int NWV, bool MIX, int NB, bool CHAIN = (NB > 1)>
__device__ __forceinline__ void codec_ffn_wave_body(const FfnWaveArgs& a, const FfnWaveArgs* blks, int tpu) {
    static_assert(NB == 1 || MIX, "the stage chain is built on the one-pass block");
    constexpr int NT = NWV * 64;
    constexpr int F = 4 * C;
    constexpr int KK1 = C / 16;            // k16 steps of the first product
    constexpr int NT1 = F / 32;            // 32-row hidden tiles
    constexpr int NOT = C / 32;            // 32-row output (channel) tiles
    constexpr int RB1 = 2 * C;             // bytes per W1 row image (64 or 128)
    constexpr int RB2 = 2 * F;             // bytes per W2 row image (256 or 512)
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
#ifndef FFN_GELU3
#define FFN_GELU3 1
#endif
    constexpr bool G3 = FFN_GELU3 && SPLIT != 3;  // three-term erfc (A&S 7.1.25) where the hidden is rounded to 16 bits anyway
#ifndef FFN_GELUQ5
#define FFN_GELUQ5 1
#endif
    // GeluQ5 (common.hpp): max(x, 0) - |x| 2^q(|x|) — 8 plain instructions, one transcendental, no packed phases: replaces the
    // A / B / C / D split below for the single-array formats (B = Horner + exp2 under the first product, D = result + convert)
    constexpr bool Q5 = FFN_GELUQ5 && SPLIT != 3;
#ifndef FFN_GELU_PK16
#define FFN_GELU_PK16 1
#endif
    // ... and for the fp16 format in PACKED fp16 (common.hpp gelu_q5_pk_*): two values per lane-instruction, 5 + 1 instead of 9 + 1
    constexpr bool PK16 = FFN_GELU_PK16 && Q5 && SPLIT == PREC_F16;
    constexpr int W_ARR = F * RB1;         // = C * RB2 = 8 C^2
    constexpr int OFF_W1 = 0, OFF_W2 = NARR * W_ARR, OFF_V = 2 * NARR * W_ARR;  // then b1[F] b2[C] gamma[C] norm_w[C] (fp32)
    constexpr int VEC_F = F + 3 * C + (MIX ? 10 * C : 0);   // ... and, MIX, the mixer's norm weight, conv bias, layer scale [C] and taps [7][C]
    constexpr int BLKB = OFF_V + VEC_F * 4;                 // bytes of ONE block's weights + vectors; block b lives at smem + b * BLKB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    auto swz1 = [](int r) { return RB1 == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    // ---- one-time: weight images into LDS (16-B chunks XOR-swizzled so every fragment read is conflict free) ----
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        const FfnWaveArgs& ab = blks[b];
        char* const wdst = smem + b * BLKB;
        for (int i = tid; i < F * (C / 8); i += NT) {
            const int r = i / (C / 8), ch = i % (C / 8);
            const int dst = r * RB1 + ((ch ^ swz1(r)) << 4);
            *reinterpret_cast<uint4*>(wdst + OFF_W1 + dst) = *reinterpret_cast<const uint4*>(ab.w1hi + (long)r * ab.ld1 + ch * 8);
            if (SPLIT == 3)
                *reinterpret_cast<uint4*>(wdst + OFF_W1 + W_ARR + dst) = *reinterpret_cast<const uint4*>(ab.w1lo + (long)r * ab.ld1 + ch * 8);
        }
        for (int i = tid; i < C * (F / 8); i += NT) {
            const int r = i / (F / 8), ch = i % (F / 8);
            const int dst = r * RB2 + (((ch & ~15) | ((ch ^ r) & 15)) << 4);
            *reinterpret_cast<uint4*>(wdst + OFF_W2 + dst) = *reinterpret_cast<const uint4*>(ab.w2hi + (long)r * F + ch * 8);
            if (SPLIT == 3)
                *reinterpret_cast<uint4*>(wdst + OFF_W2 + W_ARR + dst) = *reinterpret_cast<const uint4*>(ab.w2lo + (long)r * F + ch * 8);
        }
        float* const v = reinterpret_cast<float*>(wdst + OFF_V);
        for (int i = tid; i < F; i += NT) v[i] = ab.b1[i];
        for (int i = tid; i < C; i += NT) { v[F + i] = ab.b2[i]; v[F + C + i] = ab.gamma[i]; v[F + 2 * C + i] = ab.norm_w[i]; }
        if (MIX) {
            for (int i = tid; i < C; i += NT) { v[F + 3 * C + i] = ab.mnorm_w[i]; v[F + 4 * C + i] = ab.dw_b[i]; v[F + 5 * C + i] = ab.mgamma[i]; }
            for (int i = tid; i < 7 * C; i += NT) v[F + 6 * C + i] = ab.dw_w[i];
        }
    }
    // the block whose weights the lambdas below address: a compile-time constant for NB == 1, advanced per block by the chain
    const char* wsm = smem;
    const float *vb1, *vb2, *vga, *vnw, *vmn, *vdb, *vmg, *vdw;
    auto select_block = [&](int b) {
        wsm = smem + (NB > 1 ? b * BLKB : 0);
        vb1 = reinterpret_cast<const float*>(wsm + OFF_V);
        vb2 = vb1 + F;
        vga = vb2 + C;
        vnw = vga + C;
        vmn = vnw + C;        // mixer norm weight [C]
        vdb = vmn + C;        // conv bias [C]
        vmg = vdb + C;        // mixer layer scale [C]
        vdw = vmg + C;        // conv taps [7][C]
    };
    select_block(0);
    constexpr int RSN = C + 4;   // n-tile row stride in floats: 16 B of padding -> the 32 frames of a ds_read_b128 hit 32 different bank quads
    // per wave: the n-tile (6 halo rows + 32 frames) and, chain, the raw halo rows of blocks 1 .. NB - 1 parked for the next tile
    constexpr int TILE_F = (38 + (NB > 1 ? 6 * (NB - 1) : 0)) * RSN;
    float* const ntile = reinterpret_cast<float*>(smem + NB * BLKB) + (size_t)wave * TILE_F;
    __syncthreads();

    // fragment byte offsets (constant for the kernel)
    int w1_off[KK1];
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) w1_off[kk] = fr * RB1 + (((2 * kk + fh) ^ swz1(fr)) << 4);  // + t * 32 * RB1 (swz unchanged)
    // W2 fragment of (hidden tile t, k16 step s, channel tile ot): row 32 ot + fr, 16-B chunk c = 4 t + 2 s + fh, stored at
    // ((c & ~15) | ((c ^ row) & 15)) << 4.  With c ^ row split into its low two bits (2 s + fh, t-free) and bits 2-3 (t & 3):
    //   offset = w2_b[s] + w2_t(t) + ot * 32 * RB2,  w2_t(t) = (((t & 3) ^ ((fr >> 2) & 3)) << 6) + (t >> 2) * 256
    int w2_b[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) w2_b[s2] = fr * RB2 + (((2 * s2 + fh) ^ (fr & 3)) << 4);
    const int rsw = (fr >> 2) & 3;
    constexpr int W2LO = W_ARR;
    constexpr int NPASS = SPLIT == 3 ? 3 : 1;

    const int ntiles = (a.M + 31) / 32;
    const int wg = blockIdx.x * NWV + wave, nwg = gridDim.x * NWV;
    // tile walk: NB == 1 strides over the image (tile wg, wg + nwg, ...); the chain takes a contiguous run [seg0, seg1) in time order,
    // preceded by one warm-up tile (computed, not stored) unless the run starts where an utterance starts
    int t_first = wg, t_last = ntiles, t_step = nwg;
    if (CHAIN) {
        const int per = (ntiles + nwg - 1) / nwg;
        const int seg0 = wg * per;
        t_last = seg0 + per < ntiles ? seg0 + per : ntiles;
        t_first = seg0 < t_last && (seg0 % tpu) != 0 ? seg0 - 1 : seg0;
        t_step = 1;
    }
    const int t_store0 = CHAIN ? wg * ((ntiles + nwg - 1) / nwg) : 0;   // chain: tiles before this one are warm-up

    // channel of element e (0..7) of k16 step kk in this lane's B fragment: 16 kk + 8 fh + e
    float4 xa[KK1][2];  // raw x of the tile being prefetched / computed
    float4 hv[2];       // MIX: 8 channels of one halo frame (lanes < 6 C / 8)
    constexpr int LPF = C / 8;   // halo: lanes per frame
    const float* const xsrc = MIX ? a.xin : a.x;
    auto load_tile = [&](int wt) {
        int m = wt * 32 + fr;
        m = m < a.M ? m : a.M - 1;
        const float* xr = xsrc + a.img.at(m);
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
            xa[kk][0] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh);
            xa[kk][1] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh + 4);
        }
        if (MIX) {   // halo frame f = lane / LPF (0..5) of the tile = 6 - f rows in front of its first frame (zero pad rows at t = 0)
            const int l = lane < 6 * LPF ? lane : 0;
            const float* hr_ = xsrc + a.img.at(wt * 32) - (long)(6 - l / LPF) * a.img.ld + 8 * (l % LPF);
            hv[0] = *reinterpret_cast<const float4*>(hr_);
            hv[1] = *reinterpret_cast<const float4*>(hr_ + 4);
        }
    };
    if (t_first < t_last) load_tile(t_first);

#pragma unroll 1
    for (int wt = t_first; wt < t_last; wt += t_step) {
      float4 xo[NOT][4];   // a block's output tile in the accumulator layout (chain: the next block's input)
#pragma unroll 1
      for (int blk = 0; blk < NB; ++blk) {
        if (NB > 1) {
            select_block(blk);
            if (blk > 0) {
                // input tile = the previous block's output: accumulator layout (4-channel groups q of channel tile ot) -> B-fragment
                // layout (8 consecutive channels per lane half), the inverse of KEEPX's swap (it is an involution)
#pragma unroll
                for (int kk = 0; kk < KK1; ++kk) {
                    const float4 e4 = xo[kk / 2][2 * (kk % 2)], o4 = xo[kk / 2][2 * (kk % 2) + 1];
                    const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, ov[4] = {o4.x, o4.y, o4.z, o4.w};
                    float a0[4], a1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ev[e]), __float_as_uint(ov[e]), false, false);
                        a0[e] = __uint_as_float(r[0]); a1[e] = __uint_as_float(r[1]);
                    }
                    xa[kk][0] = make_float4(a0[0], a0[1], a0[2], a0[3]);
                    xa[kk][1] = make_float4(a1[0], a1[1], a1[2], a1[3]);
                }
                // its halo: the last six frames of this block's input on the previous tile (zeros where an utterance / the run starts)
                float* const hs = ntile + (38 + 6 * (blk - 1)) * RSN;
                const bool fresh = wt == t_first || (wt % tpu) == 0;
                const int l = lane < 6 * LPF ? lane : 0;
                const float4 h0 = *reinterpret_cast<const float4*>(hs + (l / LPF) * RSN + 8 * (l % LPF));
                const float4 h1 = *reinterpret_cast<const float4*>(hs + (l / LPF) * RSN + 8 * (l % LPF) + 4);
                hv[0] = fresh ? make_float4(0.f, 0.f, 0.f, 0.f) : h0;
                hv[1] = fresh ? make_float4(0.f, 0.f, 0.f, 0.f) : h1;
                if (fr >= 26) {   // ... and park this tile's last six input frames for the next tile (same wave: LDS operations stay in order)
#pragma unroll
                    for (int kk = 0; kk < KK1; ++kk) {
                        *reinterpret_cast<float4*>(hs + (fr - 26) * RSN + 16 * kk + 8 * fh) = xa[kk][0];
                        *reinterpret_cast<float4*>(hs + (fr - 26) * RSN + 16 * kk + 8 * fh + 4) = xa[kk][1];
                    }
                }
            }
        }
        if (MIX) {
            // ---- mixer: u = x * rstd (no affine) of the 32 + 6 frames -> LDS, conv over the seven rows, LayerScale residual ----
            float ms = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    ms = ssq4(xa[kk][h2], ms);
            ms += __shfl_xor(ms, 32, 64);
            const float mr = 1.0f / sqrtf(fmaf(ms, 1.0f / (float)C, a.eps));
            float hs = ssq4(hv[1], ssq4(hv[0], 0.f));
#pragma unroll
            for (int o = LPF >> 1; o > 0; o >>= 1) hs += __shfl_xor(hs, o, 64);
            const float hrs = 1.0f / sqrtf(fmaf(hs, 1.0f / (float)C, a.eps));
            float* own = ntile + (6 + fr) * RSN + 8 * fh;
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                *reinterpret_cast<float4*>(own + 16 * kk) = make_float4(xa[kk][0].x * mr, xa[kk][0].y * mr, xa[kk][0].z * mr, xa[kk][0].w * mr);
                *reinterpret_cast<float4*>(own + 16 * kk + 4) = make_float4(xa[kk][1].x * mr, xa[kk][1].y * mr, xa[kk][1].z * mr, xa[kk][1].w * mr);
            }
            if (lane < 6 * LPF) {
                float* hrow = ntile + (lane / LPF) * RSN + 8 * (lane % LPF);
                *reinterpret_cast<float4*>(hrow) = make_float4(hv[0].x * hrs, hv[0].y * hrs, hv[0].z * hrs, hv[0].w * hrs);
                *reinterpret_cast<float4*>(hrow + 4) = make_float4(hv[1].x * hrs, hv[1].y * hrs, hv[1].z * hrs, hv[1].w * hrs);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: written and read by this wave only
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int c = 16 * kk + 8 * fh + 4 * h2;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const float4 u4 = *reinterpret_cast<const float4*>(ntile + (fr + k) * RSN + c);
                        const float4 w4 = *reinterpret_cast<const float4*>(vdw + k * C + c);
                        acc.x = fmaf(w4.x, u4.x, acc.x); acc.y = fmaf(w4.y, u4.y, acc.y);
                        acc.z = fmaf(w4.z, u4.z, acc.z); acc.w = fmaf(w4.w, u4.w, acc.w);
                    }
                    const float4 g4 = *reinterpret_cast<const float4*>(vmn + c);
                    const float4 b4 = *reinterpret_cast<const float4*>(vdb + c);
                    const float4 m4 = *reinterpret_cast<const float4*>(vmg + c);
                    float4& xv = xa[kk][h2];
                    xv.x = fmaf(m4.x, fmaf(g4.x, acc.x, b4.x), xv.x); xv.y = fmaf(m4.y, fmaf(g4.y, acc.y, b4.y), xv.y);
                    xv.z = fmaf(m4.z, fmaf(g4.z, acc.z, b4.z), xv.z); xv.w = fmaf(m4.w, fmaf(g4.w, acc.w, b4.w), xv.w);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane's conv reads are done: the tile's rows can be reused
            // the residual add at the end of the block needs x_mid again: it waits in the (now free) n-tile instead of C / 2 registers
            // per lane (at C = 64 those registers spilled), and the epilogue reads it back directly in the accumulator layout
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                *reinterpret_cast<float4*>(own + 16 * kk) = xa[kk][0];
                *reinterpret_cast<float4*>(own + 16 * kk + 4) = xa[kk][1];
            }
        }
        // ---- RMSNorm of the lane's frame (channels split over the two lane halves) -> split bf16 B fragments ----
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                ss = ssq4(xa[kk][h2], ss);
        ss += __shfl_xor(ss, 32, 64);
        const float rstd = 1.0f / sqrtf(fmaf(ss, 1.0f / (float)C, a.eps));
        bf16x8 nh[KK1], nl[KK1];
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
            const float4 g0 = *reinterpret_cast<const float4*>(vnw + 16 * kk + 8 * fh);
            const float4 g1 = *reinterpret_cast<const float4*>(vnw + 16 * kk + 8 * fh + 4);
            const float v[8] = {xa[kk][0].x * rstd * g0.x, xa[kk][0].y * rstd * g0.y, xa[kk][0].z * rstd * g0.z, xa[kk][0].w * rstd * g0.w,
                                xa[kk][1].x * rstd * g1.x, xa[kk][1].y * rstd * g1.y, xa[kk][1].z * rstd * g1.z, xa[kk][1].w * rstd * g1.w};
            unsigned nhp[4], nlp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (SPLIT == PREC_F16) { nhp[e] = cvt_pk_f16_raw(v[2 * e], v[2 * e + 1]); nlp[e] = 0; }   // |n| <= sqrt(C) max |g|: certified at finalize
                else split_pair(v[2 * e], v[2 * e + 1], nhp[e], nlp[e]);
            }
            nh[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nhp));
            nl[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nlp));
        }
        const int m_cur = wt * 32 + fr;
#ifndef FW_KEEPX
#define FW_KEEPX 1
#endif
        // The residual add needs this tile's x again.  Re-reading it in the epilogue made the C = 32 / 64 kernels move 3 |x| per
        // launch (read, re-read, write) at 4.0-4.6 TB/s — they were HBM-bound, which is why neither fewer VALU instructions nor more
        // waves moved them (profiles/r02j_*, r02f_*).  The tile stays in registers instead (C / 2 per lane) and is brought from the
        // B-fragment layout (8 consecutive channels per lane half) into the accumulator layout (4-channel groups) by one
        // v_permlane32_swap per register in the epilogue.
        constexpr bool KEEPX = !MIX && FW_KEEPX && (SPLIT != 3 || C == 32);   // (C = 64 at split-bf16 is out of registers: it re-reads)
        float4 xk[KK1][2];
        if (KEEPX) {
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) { xk[kk][0] = xa[kk][0]; xk[kk][1] = xa[kk][1]; }
        }
        if ((NB == 1 || blk == NB - 1) && wt + t_step < t_last) load_tile(wt + t_step);  // next tile's x arrives under this tile's MFMA / GELU work

        floatx16 acc2[NOT];
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;

        // ---- software pipeline over the hidden tiles --------------------------------------------------------------
        // On gfx950 v_pk_*_f32 instructions do not run next to MFMAs (they serialise with the matrix pipe, measured in
        // tools/ubench/mfma_valu.hip) while plain VALU / transcendental instructions do.  One step therefore handles three
        // hidden tiles at once: the MFMAs of the first product of tile t+1 and of the second product of tile t-1 carry the
        // non-packed half of tile t's GELU + split in their shadow (rcp / exp2 under the former; the final fma, the bf16
        // splits and the lane swaps under the latter), the packed polynomial runs between the two MFMA groups.
        // sched_barrier(0) pins the source order, which IS the schedule: one MFMA, then its share of the VALU work.
        struct Frags { bf16x8 h[2], l[2]; };   // B fragments (two k16 steps) of one GELU'd hidden tile
        auto mfma3 = [&](floatx16& acc, const bf16x8 (&w)[2], const bf16x8& bh, const bf16x8& bl, int pass) {
            // pass order: the two cross terms first, hi . hi last (as before)
            if (SPLIT == 3 && pass == 0) acc = mfma16<SPLIT>(w[1], bh, acc);
            else if (SPLIT == 3 && pass == 1) acc = mfma16<SPLIT>(w[0], bl, acc);
            else acc = mfma16<SPLIT>(w[0], bh, acc);
        };
        // accumulator row r of hidden tile t is hidden unit 32 t + (r & 3) + 8 (r >> 2) + 4 fh
        auto bias_init = [&](floatx16& acc, int t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(vb1 + 32 * t + 8 * q + 4 * fh);
                acc[4 * q + 0] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
            }
        };
        auto step = [&](auto do_p1, auto do_p2, int t, const floatx16& h, floatx16& accn, const Frags& prev, Frags& cur) {
            constexpr bool P1 = decltype(do_p1)::value, P2 = decltype(do_p2)::value;
            // fragment reads for the first MFMA group of each product, and the tile's bias
            bf16x8 w1f[2][2], w2f[2][2];
            const int w1t = (t + 1) * 32 * RB1;
            const int w2t = (((t - 1) & 3) ^ rsw) * 64 + ((t - 1) >> 2) * 256;
            if (P1) {
                w1f[0][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + w1t + w1_off[0]);
                if (SPLIT == 3) w1f[0][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + W_ARR + w1t + w1_off[0]);
            }
            if (P1) bias_init(accn, t + 1);  // the accumulators start from b1: no separate bias add
            // ---- A (packed): exponent argument ----
            f32x2 u[8], ea[8];
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                u[pr].x = h[2 * pr]; u[pr].y = h[2 * pr + 1];
                if constexpr (!Q5) ea[pr] = (u[pr] * u[pr]) * (-0.5f * 1.4426950408889634f);  // exp(-z^2) = exp2(-x^2/2 log2 e)
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- B: first product of tile t+1 || t = 1 / (1 + p z), e = exp2(.) of the 16 values ----
            f32x2 tt[8], ee[8];
            unsigned hp[8], axp[8], eep[8];   // PK16: the tile as packed fp16 pairs, |.| of them, 2^q(|.|)
            auto trans = [&](int v) {
                if constexpr (PK16) {
                    if (!(v & 1)) gelu_q5_pk_front(u[v >> 1].x, u[v >> 1].y, hp[v >> 1], axp[v >> 1], eep[v >> 1]);
                    return;
                }
                const float xv = (v & 1) ? u[v >> 1].y : u[v >> 1].x;
                if constexpr (Q5) {
                    const float ax = fabsf(xv);
                    float q = fmaf(GeluQ5::Q5, ax, GeluQ5::Q4);
                    q = fmaf(q, ax, GeluQ5::Q3);
                    q = fmaf(q, ax, GeluQ5::Q2);
                    q = fmaf(q, ax, GeluQ5::Q1);
                    q = fmaf(q, ax, GeluQ5::Q0);
                    const float ev = __builtin_amdgcn_exp2f(q);
                    if (v & 1) ee[v >> 1].y = ev; else ee[v >> 1].x = ev;
                } else {
                    const float av = (v & 1) ? ea[v >> 1].y : ea[v >> 1].x;
                    const float tv = fast_rcp(fmaf(fabsf(xv), G3 ? Gelu3::P : 0.3275911f * 0.70710678118654752f, 1.0f));
                    const float ev = __builtin_amdgcn_exp2f(av);
                    if (v & 1) { tt[v >> 1].y = tv; ee[v >> 1].y = ev; } else { tt[v >> 1].x = tv; ee[v >> 1].x = ev; }
                }
            };
            if (P1) {
                constexpr int NMB = KK1 * NPASS;
#pragma unroll
                for (int kk = 0; kk < KK1; ++kk) {
                    if (kk + 1 < KK1) {
                        w1f[(kk + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + w1t + w1_off[kk + 1]);
                        if (SPLIT == 3) w1f[(kk + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + W_ARR + w1t + w1_off[kk + 1]);
                    } else if (P2) {
                        w2f[0][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + w2_b[0] + w2t);
                        if (SPLIT == 3) w2f[0][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + W2LO + w2_b[0] + w2t);
                    }
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        const int j = kk * NPASS + ps;
                        __builtin_amdgcn_sched_barrier(0);
                        mfma3(accn, w1f[kk & 1], nh[kk], nl[kk], ps);
#pragma unroll
                        for (int v = 16 * j / NMB; v < 16 * (j + 1) / NMB; ++v) trans(v);
                    }
                }
            } else {
                if (P2) {
                    w2f[0][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + w2_b[0] + w2t);
                    if (SPLIT == 3) w2f[0][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + W2LO + w2_b[0] + w2t);
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) trans(v);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- C (packed): erfc polynomial, erf, x / 2 ----
            f32x2 uu[8], hx[8];
            if constexpr (!Q5)
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const f32x2 t2 = tt[pr];
                const f32x2 poly = G3 ? t2 * (Gelu3::A1 + t2 * (Gelu3::A2 + t2 * Gelu3::A3))
                                      : t2 * (0.254829592f + t2 * (-0.284496736f + t2 * (1.421413741f + t2 * (-1.453152027f + t2 * 1.061405429f))));
                uu[pr] = 1.0f - poly * ee[pr];  // erf(z)
                hx[pr] = 0.5f * u[pr];
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- D: second product of tile t-1 || gelu = x/2 + |x/2| erf, bf16 split, lane swap -> fragments of tile t ----
            unsigned hiP[8], loP[8];
            auto finish = [&](int pr) {
                if constexpr (PK16) {
                    hiP[pr] = gelu_q5_pk_back(hp[pr], axp[pr], eep[pr]);
                    return;
                }
                float rx, ry;
                if constexpr (Q5) {
                    rx = fmaf(-fabsf(u[pr].x), ee[pr].x, relu_f(u[pr].x));
                    ry = fmaf(-fabsf(u[pr].y), ee[pr].y, relu_f(u[pr].y));
                } else {
                    rx = fmaf(fabsf(hx[pr].x), uu[pr].x, hx[pr].x);
                    ry = fmaf(fabsf(hx[pr].y), uu[pr].y, hx[pr].y);
                }
                hiP[pr] = SPLIT == PREC_F16 ? cvt_pk_f16_satpos(rx, ry) : cvt_pk_bf16(rx, ry);
                if (SPLIT == 3) {
                    f32x2 rr, hf;
                    rr.x = rx; rr.y = ry;
                    hf.x = __uint_as_float(hiP[pr] << 16); hf.y = __uint_as_float(hiP[pr] & 0xffff0000u);
                    const f32x2 lo2 = rr - hf;
                    loP[pr] = cvt_pk_bf16(lo2.x, lo2.y);
                }
            };
            auto swap_half = [&](int s2) {
                // lane half 0 needs hidden 0..7 of the k16 step, half 1 hidden 8..15: swap upper half of X with lower half of Y
                unsigned fhh[4], fll[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto rh = __builtin_amdgcn_permlane32_swap(hiP[4 * s2 + e], hiP[4 * s2 + 2 + e], false, false);
                    fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                    if (SPLIT == 3) {
                        auto rl = __builtin_amdgcn_permlane32_swap(loP[4 * s2 + e], loP[4 * s2 + 2 + e], false, false);
                        fll[e] = rl[0]; fll[2 + e] = rl[1];
                    }
                }
                cur.h[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                if (SPLIT == 3) cur.l[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fll));
            };
            if (P2) {
                constexpr int ND = 2 * NOT * NPASS;
#pragma unroll
                for (int g = 0; g < 2 * NOT; ++g) {
                    const int s2 = g / NOT, ot = g % NOT;
                    if (g + 1 < 2 * NOT) {
                        const int s3 = (g + 1) / NOT, ot3 = (g + 1) % NOT;
                        const char* nx = wsm + OFF_W2 + w2_b[s3] + w2t + ot3 * 32 * RB2;
                        w2f[(g + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(nx);
                        if (SPLIT == 3) w2f[(g + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(nx + W2LO);
                    }
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        const int j = g * NPASS + ps;
                        __builtin_amdgcn_sched_barrier(0);
                        mfma3(acc2[ot], w2f[g & 1], prev.h[s2], prev.l[s2], ps);
#pragma unroll
                        for (int pr = 8 * j / ND; pr < 8 * (j + 1) / ND; ++pr) {
                            finish(pr);
                            if (pr == 3) swap_half(0);
                            if (pr == 7) swap_half(1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) finish(pr);
                swap_half(0);
                swap_half(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        floatx16 hA, hB;
        Frags fX, fY;
        {   // pipeline fill: first product of hidden tile 0
            bias_init(hA, 0);
            bf16x8 w1f[2][2];
            w1f[0][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + w1_off[0]);
            if (SPLIT == 3) w1f[0][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + W_ARR + w1_off[0]);
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                if (kk + 1 < KK1) {
                    w1f[(kk + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + w1_off[kk + 1]);
                    if (SPLIT == 3) w1f[(kk + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W1 + W_ARR + w1_off[kk + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) mfma3(hA, w1f[kk & 1], nh[kk], nl[kk], ps);
            }
        }
        step(T_{}, F_{}, 0, hA, hB, fY, fX);  // tile 0: GELU || first product of tile 1
        static_assert(NT1 >= 4 && NT1 % 2 == 0, "pipeline assumes an even number (>= 4) of hidden tiles");
#pragma unroll 1
        for (int t = 1; t < NT1 - 1; t += 2) {  // two steps per trip: the accumulators and fragment sets swap roles
            step(T_{}, T_{}, t, hB, hA, fX, fY);
            step(T_{}, T_{}, t + 1, hA, hB, fY, fX);
        }
        step(F_{}, T_{}, NT1 - 1, hB, hA, fX, fY);
        {   // pipeline drain: second product of the last hidden tile
            const int w2t = (((NT1 - 1) & 3) ^ rsw) * 64 + ((NT1 - 1) >> 2) * 256;
            bf16x8 w2f[2][2];
            w2f[0][0] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + w2_b[0] + w2t);
            if (SPLIT == 3) w2f[0][1] = *reinterpret_cast<const bf16x8*>(wsm + OFF_W2 + W2LO + w2_b[0] + w2t);
#pragma unroll
            for (int g = 0; g < 2 * NOT; ++g) {
                const int s2 = g / NOT, ot = g % NOT;
                if (g + 1 < 2 * NOT) {
                    const int s3 = (g + 1) / NOT, ot3 = (g + 1) % NOT;
                    const char* nx = wsm + OFF_W2 + w2_b[s3] + w2t + ot3 * 32 * RB2;
                    w2f[(g + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(nx);
                    if (SPLIT == 3) w2f[(g + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(nx + W2LO);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) mfma3(acc2[ot], w2f[g & 1], fY.h[s2], fY.l[s2], ps);
            }
        }
        // ---- epilogue: x[frame][c] += gamma[c] (out + b2[c]); channel(r) = 32 ot + (r & 3) + 8 (r >> 2) + 4 fh -------
        if (m_cur < a.M) {
            float* xr = a.x + a.img.at(m_cur);
            const bool store = blk == NB - 1 && wt >= t_store0;   // chain: the last block of a tile that is not warm-up
            if (KEEPX) {
                // xk[kk][h2] holds channels 16 kk + 8 fh + 4 h2 + (0..3); the accumulator rows 4 q .. 4 q + 3 of channel tile ot are
                // channels 32 ot + 8 q + 4 fh + (0..3) = 16 kk' + 8 (q & 1) + 4 fh + (0..3), kk' = 2 ot + q / 2.  Swapping the upper lane
                // half of xk[kk'][0] with the lower lane half of xk[kk'][1] leaves {fh 0: +0..3, fh 1: +4..7} in the first and
                // {fh 0: +8..11, fh 1: +12..15} in the second: exactly q even / q odd for both lane halves.
#pragma unroll
                for (int kk = 0; kk < KK1; ++kk) {
                    float xe[4], xo2[4];
                    const float a0[4] = {xk[kk][0].x, xk[kk][0].y, xk[kk][0].z, xk[kk][0].w};
                    const float a1[4] = {xk[kk][1].x, xk[kk][1].y, xk[kk][1].z, xk[kk][1].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0[e]), __float_as_uint(a1[e]), false, false);
                        xe[e] = __uint_as_float(r[0]); xo2[e] = __uint_as_float(r[1]);
                    }
                    xo[kk / 2][2 * (kk % 2)] = make_float4(xe[0], xe[1], xe[2], xe[3]);
                    xo[kk / 2][2 * (kk % 2) + 1] = make_float4(xo2[0], xo2[1], xo2[2], xo2[3]);
                }
            } else if (MIX) {   // x_mid from the wave's n-tile (written there after the conv)
                const float* xm = ntile + (6 + fr) * RSN + 4 * fh;
#pragma unroll
                for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xo[ot][q] = *reinterpret_cast<const float4*>(xm + 32 * ot + 8 * q);
            } else {
#pragma unroll
                for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xo[ot][q] = *reinterpret_cast<const float4*>(xr + 32 * ot + 8 * q + 4 * fh);
            }
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * ot + 8 * q + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(vb2 + c0);
                    const float4 gv = *reinterpret_cast<const float4*>(vga + c0);
                    float4 o = xo[ot][q];
                    o.x = fmaf(gv.x, acc2[ot][4 * q + 0] + bv.x, o.x);
                    o.y = fmaf(gv.y, acc2[ot][4 * q + 1] + bv.y, o.y);
                    o.z = fmaf(gv.z, acc2[ot][4 * q + 2] + bv.z, o.z);
                    o.w = fmaf(gv.w, acc2[ot][4 * q + 3] + bv.w, o.w);
                    if (store) *reinterpret_cast<float4*>(xr + c0) = o;
                    if (NB > 1) xo[ot][q] = o;
                }
        }
      }
    }
}

template <int C, int SPLIT, int NWV, bool MIX = false>
__global__ __launch_bounds__(NWV * 64, (C == 64 && SPLIT != 3 && NWV == 4) ? FW_MINW64 : (C == 32 && SPLIT != 3) ? FW_MINW32 : 1) void codec_ffn_wave_kernel(FfnWaveArgs a) {
    codec_ffn_wave_body<C, SPLIT, NWV, MIX, 1>(a, &a, 0);
}

// the stage chain: NB one-pass blocks per tile (C = 32: all three blocks' fp16 weights are LDS-resident, 48 KiB)
template <int C, int SPLIT, int NWV, int NB>
__global__ __launch_bounds__(NWV * 64) void codec_chain_wave_kernel(FfnChainArgs c) {
    codec_ffn_wave_body<C, SPLIT, NWV, true, NB, true>(c.b[0], c.b, c.tpu);
}

#ifndef FW_NWV32
#define FW_NWV32 8
#endif
template <int C, int SPLIT, bool MIX = false>
static hipError_t ffn_wave_go(const FfnWaveArgs& a, hipStream_t st) {
    constexpr int NWV = (C == 32 && SPLIT != 3) ? FW_NWV32 : (C == 64 && SPLIT != 3) ? FW_NWV64 : 8;
    constexpr size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * 2 * (8 * C * C) + (size_t)(4 * C + 3 * C) * 4 +
                           (MIX ? (size_t)(10 * C + NWV * 38 * (C + 4)) * 4 : 0);
    static_assert(lds <= 160 * 1024, "weights must fit LDS");
    auto kern = codec_ffn_wave_kernel<C, SPLIT, NWV, MIX>;
    static DevOnce once;
    int cus = 256;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }, &cus);
    if (e != hipSuccess) return e;
    if (!(g_persist_mask & 2)) cus = once.real_cus();   // (A/B: which persistent kernels the throughput-mode grid cap applies to)
    const int ntiles = (a.M + 31) / 32;
    // persistent: as many workgroups as fit (LDS, and 32 waves per CU), each wave walks tiles
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 32 / NWV ? 32 / NWV : per_cu;
    per_cu = per_cu > (NWV == 8 ? 2 : 4) ? (NWV == 8 ? 2 : 4) : per_cu;
    int grid = (ntiles + NWV - 1) / NWV;
    grid = grid < cus * per_cu ? grid : cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, a);
    return hipGetLastError();
}

// C in {32, 64}; w1 [F][ld1] (ld1 >= C, K possibly zero-padded), w2 [>= C][F]
hipError_t launch_codec_ffn_wave(float* x, RowMap img, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo, int ld1,
                                 const float* b1, const bf16_t* w2hi, const bf16_t* w2lo, const float* b2, const float* gamma,
                                 int M, int C, int F, float eps, int split, hipStream_t st) {
    if (!(C == 32 || C == 64) || F != 4 * C || img.ld % 4 || img.off % 4 || ld1 % 8 || (img.rpb && img.bstride % 4)) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    FfnWaveArgs a{x, img, norm_w, w1hi, w1lo, ld1, b1, w2hi, w2lo, b2, gamma, M, eps};
    ProfScope ps(st, C == 64 ? "codec_ffn_wave<64>" : "codec_ffn_wave<32>", 4.0 * M * (double)C * F, 8.0 * M * C + 8.0 * (double)C * F);
    if (C == 64) return split == 3 ? ffn_wave_go<64, 3>(a, st) : split == PREC_F16 ? ffn_wave_go<64, 2>(a, st) : ffn_wave_go<64, 1>(a, st);
    return split == 3 ? ffn_wave_go<32, 3>(a, st) : split == PREC_F16 ? ffn_wave_go<32, 2>(a, st) : ffn_wave_go<32, 1>(a, st);
}

// the whole block (mixer + FFN) in one pass for C in {32, 64} at the single-array formats (and C = 32 at split-bf16):
// xout = block(xin); xin != xout, both images with zero pad frames; needs T % 32 == 0 (img.rpb) and 7 conv taps
bool codec_block_wave_ok(int C, int F, int K, int T, int split) {
    return (C == 32 || C == 64) && F == 4 * C && K == 7 && T > 0 && T % 32 == 0 && (split != 3 || C == 32);
}
hipError_t launch_codec_block_wave(const float* xin, float* xout, RowMap img, const float* mnorm_w, const float* dw_w, const float* dw_b,
                                   const float* mgamma, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo, int ld1,
                                   const float* b1, const bf16_t* w2hi, const bf16_t* w2lo, const float* b2, const float* gamma, int M,
                                   int C, int F, int K, float eps, int split, hipStream_t st) {
    if (!codec_block_wave_ok(C, F, K, img.rpb, split) || xin == xout || img.ld != C || img.off % 4 || ld1 % 8 || img.bstride % 4 ||
        M % 32)
        return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    FfnWaveArgs a{xout, img, norm_w, w1hi, w1lo, ld1, b1, w2hi, w2lo, b2, gamma, M, eps, xin, mnorm_w, dw_w, dw_b, mgamma};
    ProfScope ps(st, C == 64 ? "codec_block_wave<64>" : "codec_block_wave<32>", 4.0 * M * (double)C * F + 2.0 * M * C * (K + 4),
                 8.0 * M * C + 8.0 * (double)C * F);
    if (C == 64) return split == PREC_F16 ? ffn_wave_go<64, 2, true>(a, st) : ffn_wave_go<64, 1, true>(a, st);
    return split == 3 ? ffn_wave_go<32, 3, true>(a, st) : split == PREC_F16 ? ffn_wave_go<32, 2, true>(a, st) : ffn_wave_go<32, 1, true>(a, st);
}

// ---- stage chain: all NB blocks of a C = 32 stage in one launch (single-array operand formats) ---------------------------------------
#ifndef FW_CHAIN_NWV
#define FW_CHAIN_NWV 12   // waves per workgroup = per CU: 3 x 18.1 KiB of weights + vectors and 7.9 KiB of tiles per wave -> 149 KiB at 12
#endif
bool codec_chain_wave_ok(int C, int F, int K, int T, int split, int nb) {
    return C == 32 && nb >= 1 && nb <= 3 && split != 3 && codec_block_wave_ok(C, F, K, T, split);
}
template <int C, int SPLIT, int NB>
static hipError_t chain_wave_go(const FfnChainArgs& c, hipStream_t st) {
    constexpr int NWV = FW_CHAIN_NWV, F = 4 * C;
    constexpr size_t blkb = (size_t)2 * (8 * C * C) + (size_t)(F + 3 * C + 10 * C) * 4;
    constexpr size_t lds = NB * blkb + (size_t)NWV * (38 + 6 * (NB - 1)) * (C + 4) * 4;
    static_assert(lds <= 160 * 1024, "chain: weights + tiles must fit LDS");
    auto kern = codec_chain_wave_kernel<C, SPLIT, NWV, NB>;
    static DevOnce once;
    int cus = 256;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }, &cus);
    if (e != hipSuccess) return e;
    if (!(g_persist_mask & 2)) cus = once.real_cus();
    const int ntiles = c.b[0].M / 32;
    int grid = (ntiles + NWV - 1) / NWV;
    grid = grid < cus ? grid : cus;   // persistent: one workgroup per CU, every wave one contiguous run of tiles
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, c);
    return hipGetLastError();
}
// xout = block[nb - 1](... block[0](xin)); xin != xout, both images with zero pad frames; blocks: {mixer norm, taps, conv bias, mixer
// layer scale, FFN norm, W1, b1, W2, b2, FFN layer scale} each
hipError_t launch_codec_chain_wave(const float* xin, float* xout, RowMap img, const CodecChainBlock* blocks, int nb, int M, int C, int F, int K,
                                   float eps, int split, hipStream_t st) {
    if (!codec_chain_wave_ok(C, F, K, img.rpb, split, nb) || xin == xout || img.ld != C || img.off % 4 || img.bstride % 4 || M % 32)
        return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    FfnChainArgs c{};
    for (int i = 0; i < nb; ++i) {
        const CodecChainBlock& b = blocks[i];
        if (b.ld1 % 8) return hipErrorInvalidValue;
        c.b[i] = FfnWaveArgs{xout, img, b.norm_w, b.w1, nullptr, b.ld1, b.b1, b.w2, nullptr, b.b2, b.gamma, M, eps, xin, b.mnorm_w, b.dw_w, b.dw_b, b.mgamma};
    }
    c.tpu = img.rpb / 32;
    ProfScope ps(st, "codec_chain_wave<32>", nb * (4.0 * M * (double)C * F + 2.0 * M * C * (K + 4)), 8.0 * M * C + nb * 8.0 * (double)C * F);
    if (nb == 1) return split == PREC_F16 ? chain_wave_go<32, 2, 1>(c, st) : chain_wave_go<32, 1, 1>(c, st);
    if (nb == 2) return split == PREC_F16 ? chain_wave_go<32, 2, 2>(c, st) : chain_wave_go<32, 1, 2>(c, st);
    return split == PREC_F16 ? chain_wave_go<32, 2, 3>(c, st) : chain_wave_go<32, 1, 3>(c, st);
}
