// Fused codec FFN for the stages whose weights do NOT fit in LDS (C = 128, 256), gfx950:
//     x += ffn_gamma * ( W2 . gelu( W1 . RMSNorm(x; g, eps) + b1 ) + b2 )
//
// Same wave-level structure as codec_ffn_wave.hip — each wave owns 32 frames, both products are computed transposed
// (lane = frame), the normalised input and the 4C-wide hidden live in registers, v_permlane32_swap turns GELU'd
// accumulators into the B fragments of the second product — but the weights stream through an LDS ring:
//   slot(t) = [ W1 rows 32t..32t+32 (32 x C) | W2 columns 32t..32t+32 (C x 32, from a tile-major repack) ], hi and lo,
// i.e. exactly what hidden tile t needs.  All waves of the workgroup consume slot t for their own frames, so the
// ring is joined by one s_barrier per hidden tile (gemm3's protocol: own DMA pieces landed -> barrier -> refill the
// slot everybody just left).  The workgroup is persistent (walks passes of NW x 32 frames) and the ring keeps
// running across passes.  Per pass the L2 -> LDS weight traffic is 32 C^2 bytes for NW x 32 frames: 4x (C = 128) less
// per frame than codec_ffn_kernel, and nothing but x itself touches HBM (the unfused C = 256 path moved the 4C-wide
// hidden through HBM twice).
#include "gemm3.hpp"
#include "kernels.hpp"
#include "prof.hpp"

typedef float f32x2s __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pair in one v_cvt_pk_bf16_f32; split_pair also returns the packed bf16 of the two residuals
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    f32x2s v;
    v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16(a, b);
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

#ifdef GELU_SCALAR
__device__ __forceinline__ float gelu1_(float x) {
    const float t = fast_rcp(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));
    const float u = fmaf(-poly, e, 1.0f);
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), u, hx);
}
#endif
__device__ __forceinline__ f32x2s gelu2s(f32x2s x) {
#ifdef GELU_SCALAR
    { f32x2s r; r.x = gelu1_(x.x); r.y = gelu1_(x.y); return r; }
#endif
    // exact-erf GELU, A&S 7.1.26 (|erf error| <= 1.5e-7): erfc(z) = t (a1 + t (a2 + ...)) exp(-z^2), t = 1 / (1 + p z),
    // z = |x| / sqrt 2.  gelu(x) = x/2 (1 + erf(x / sqrt 2)) = x/2 + |x|/2 (1 - erfc(z)): no compare / select, the |.| are
    // free source modifiers, the polynomial runs on v_pk_* ops (two values per instruction).
    f32x2s t;
    t.x = fast_rcp(fmaf(fabsf(x.x), 0.3275911f * 0.70710678118654752f, 1.0f));
    t.y = fast_rcp(fmaf(fabsf(x.y), 0.3275911f * 0.70710678118654752f, 1.0f));
    const f32x2s poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2s ea = (x * x) * (-0.5f * 1.4426950408889634f);  // exp(-z^2) = exp2(-x^2/2 log2 e)
    f32x2s e;
    e.x = __builtin_amdgcn_exp2f(ea.x);
    e.y = __builtin_amdgcn_exp2f(ea.y);
    const f32x2s u = 1.0f - poly * e;  // erf(z)
    const f32x2s hx = 0.5f * x;
    f32x2s r;
    r.x = fmaf(fabsf(hx.x), u.x, hx.x);
    r.y = fmaf(fabsf(hx.y), u.y, hx.y);
    return r;
}

struct FfnStreamArgs {
    float* x;
    RowMap img;
    const float* norm_w;   // [C]
    const bf16_t* w1hi;    // [F][C]
    const bf16_t* w1lo;
    const float* b1;       // [F]
    const bf16_t* w2thi;   // tile-major [F/32][C][32]: w2t[t][c][k] = W2[c][32 t + k]
    const bf16_t* w2tlo;
    const float* b2;       // [C]
    const float* gamma;    // [C]
    int M;
    float eps;
};

template <int C, int SPLIT, int NW, int S, int TPB>
__global__ __launch_bounds__(NW * 64) void codec_ffn_stream_kernel(FfnStreamArgs a) {
    constexpr int F = 4 * C;
    constexpr int KK1 = C / 16;             // k16 steps of the first product
    constexpr int NT1 = F / 32;             // hidden tiles
    constexpr int NOT = C / 32;             // output (channel) tiles
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
    constexpr int RB1 = 2 * C;              // bytes per W1 row (256 / 512)
    constexpr int CPR1 = RB1 / 16;          // 16-B chunks per W1 row (16 / 32)
    constexpr int W1T = 32 * RB1;           // bytes of one W1 tile image per array (= 64 C)
    constexpr int W2T = C * 64;             // bytes of one W2 tile image per array (C rows x 64 B)
    constexpr int SUB = NARR * (W1T + W2T); // one hidden tile's weights
    constexpr int SLOT = TPB * SUB;         // a ring slot holds TPB consecutive hidden tiles (one barrier per slot)
    constexpr int NST = NT1 / TPB;          // slots per pass
    constexpr int PIECES = SLOT / 1024;     // DMA pieces per slot
    constexpr int PW = PIECES / NW;         // per wave
    constexpr int HALF = NARR * (W1T / 1024);  // pieces [0, HALF) are W1, [HALF, PIECES) W2
    constexpr int OFF_V = S * SLOT;         // b1[F] b2[C] gamma[C] norm_w[C] (fp32)
    static_assert(PW * NW == PIECES, "DMA pieces must divide over the waves");
    static_assert((S - 2) * PW <= 63 && S >= 2, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)SM_LPTR(smem);

    float* vb1 = reinterpret_cast<float*>(smem + OFF_V);
    float* vb2 = vb1 + F;
    float* vga = vb2 + C;
    float* vnw = vga + C;
    for (int i = tid; i < F; i += NW * 64) vb1[i] = a.b1[i];
    for (int i = tid; i < C; i += NW * 64) { vb2[i] = a.b2[i]; vga[i] = a.gamma[i]; vnw[i] = a.norm_w[i]; }

    // ---- this wave's DMA pieces: source pointer at hidden tile 0 and LDS offset inside a slot -------------------
    const bf16_t* src[PW];
    unsigned dst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int qq = wave * PW + i;  // wave-uniform
        const int u = qq / (SUB / 1024), q = qq % (SUB / 1024);  // hidden tile inside the slot, piece inside the tile
        if (q < HALF) {               // W1 tile: rows of RB1 bytes, 1024 / RB1 rows per piece
            const int arr = q / (W1T / 1024), j = q % (W1T / 1024);
            const int r = j * (1024 / RB1) + lane / CPR1, pos = lane % CPR1;
            const int c = (pos & ~15) | ((pos ^ r) & 15);
            src[i] = (arr ? a.w1lo : a.w1hi) + (long)r * C + c * 8 + (long)u * 32 * C;
            dst[i] = u * SUB + arr * W1T + j * 1024;
        } else {                      // W2 tile: 64-B rows, 16 rows per piece
            const int q2 = q - HALF;
            const int arr = q2 / (W2T / 1024), j = q2 % (W2T / 1024);
            const int r = j * 16 + (lane >> 2), pos = lane & 3;
            const int c = pos ^ ((r >> 2) & 3);
            src[i] = (arr ? a.w2tlo : a.w2thi) + (long)r * 32 + c * 8 + (long)u * 32 * C;
            dst[i] = u * SUB + NARR * W1T + arr * W2T + j * 1024;
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
    // hidden tile t of either matrix starts 32 * C elements after tile t - 1; slot index i covers tiles TPB * (i % NST) ..
    auto issue = [&](int i) {
        const int t = (i % NST) * TPB;
        const unsigned st = lds0 + (unsigned)((i % S) * SLOT);
#pragma unroll
        for (int p = 0; p < PW; ++p)
            dma16(src[p] + (long)t * 32 * C, st + (unsigned)__builtin_amdgcn_readfirstlane((int)dst[p]));
    };

    const int npass_total = (a.M + NW * 32 - 1) / (NW * 32);
    const int my_passes = blockIdx.x < npass_total ? (npass_total - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int total = my_passes * NST;  // ring slots this workgroup will consume
#pragma unroll 1
    for (int s = 0; s < S - 1; ++s)
        if (s < total) issue(s);

    // fragment byte offsets inside a slot
    int w1_off[KK1];
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) {
        const int c = 2 * kk + fh;
        w1_off[kk] = fr * RB1 + (((c & ~15) | ((c ^ fr) & 15)) << 4);
    }
    int w2_off[NOT][2];
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = 32 * ot + fr;
            w2_off[ot][s] = NARR * W1T + row * 64 + (((2 * s + fh) ^ ((row >> 2) & 3)) << 4);
        }

    int it = 0;  // global hidden-tile counter (ring position)
#pragma unroll 1
    for (int p = 0; p < my_passes; ++p) {
        const int pass = blockIdx.x + p * gridDim.x;
        const int m_cur = (pass * NW + wave) * 32 + fr;
        // ---- x -> RMSNorm -> split bf16 B fragments (lane = frame, channels 16 kk + 8 fh + e) -------------------
        bf16x8 nh[KK1], nl[KK1];
        {
            const int m = m_cur < a.M ? m_cur : a.M - 1;
            const float* xr = a.x + a.img.at(m);
            float4 xa[KK1][2];
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                xa[kk][0] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh);
                xa[kk][1] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh + 4);
            }
            float ss = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    ss += xa[kk][h2].x * xa[kk][h2].x + xa[kk][h2].y * xa[kk][h2].y + xa[kk][h2].z * xa[kk][h2].z + xa[kk][h2].w * xa[kk][h2].w;
            ss += __shfl_xor(ss, 32, 64);
            const float rstd = 1.0f / sqrtf(ss / (float)C + a.eps);
            if (p == 0) __syncthreads();  // vnw (and the other LDS vectors) written above are visible
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                const float4 g0 = *reinterpret_cast<const float4*>(vnw + 16 * kk + 8 * fh);
                const float4 g1 = *reinterpret_cast<const float4*>(vnw + 16 * kk + 8 * fh + 4);
                const float v[8] = {xa[kk][0].x * rstd * g0.x, xa[kk][0].y * rstd * g0.y, xa[kk][0].z * rstd * g0.z, xa[kk][0].w * rstd * g0.w,
                                    xa[kk][1].x * rstd * g1.x, xa[kk][1].y * rstd * g1.y, xa[kk][1].z * rstd * g1.z, xa[kk][1].w * rstd * g1.w};
                unsigned nhp[4], nlp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair(v[2 * e], v[2 * e + 1], nhp[e], nlp[e]);
                nh[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nhp));
                nl[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nlp));
            }
        }
        floatx16 acc2[NOT];
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;

#pragma unroll 1
        for (int t = 0; t < NT1; ++t) {
          if (t % TPB == 0) {
            // this wave's pieces of slot `it` have landed: younger = the (S-2) tiles issued after it (none near the end)
            if (it + S - 1 <= total)
                wait_vmcnt<(S - 2) * PW>();
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody left tile it-1 -> its slot is free
            if (it + S - 1 < total) issue(it + S - 1);
            ++it;
          }
            const char* sl = smem + ((it - 1) % S) * SLOT + (t % TPB) * SUB;

            // ---- H^T tile: hidden rows 32 t .. +32 x this wave's 32 frames ----------------------------------------------
            floatx16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
            // fragment reads run one k16 step ahead of the MFMAs that consume them (the compiler keeps this order and counts
            // lgkmcnt, so the LDS latency of step kk+1 hides under the three MFMAs of step kk)
            bf16x8 w1f[2][2];  // [buffer kk & 1][hi | lo]: compile-time ping-pong, no register copies
            w1f[0][0] = *reinterpret_cast<const bf16x8*>(sl + w1_off[0]);
            if (SPLIT == 3) w1f[0][1] = *reinterpret_cast<const bf16x8*>(sl + W1T + w1_off[0]);
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                if (kk + 1 < KK1) {
                    w1f[(kk + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(sl + w1_off[kk + 1]);
                    if (SPLIT == 3) w1f[(kk + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(sl + W1T + w1_off[kk + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs below
                if (SPLIT == 3) {
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[kk & 1][1], nh[kk], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[kk & 1][0], nl[kk], acc1, 0, 0, 0);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[kk & 1][0], nh[kk], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // W2 fragments are read one (k16 step, channel tile) pair ahead as well; the first pair is requested before the
            // GELU so it lands under it
            auto w2_addr = [&](int ot, int s) { return sl + w2_off[ot][s]; };
            constexpr int W2LO = W2T;
            bf16x8 w2f[2][2];  // [buffer][hi | lo], buffer = (s * NOT + ot) & 1
            w2f[0][0] = *reinterpret_cast<const bf16x8*>(w2_addr(0, 0));
            if (SPLIT == 3) w2f[0][1] = *reinterpret_cast<const bf16x8*>(w2_addr(0, 0) + W2LO);
            __builtin_amdgcn_sched_barrier(0);
            // ---- bias + GELU in place: row(r) = hidden 32 t + (r & 3) + 8 (r >> 2) + 4 fh -----------------------------
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(vb1 + 32 * t + 8 * q + 4 * fh);
                f32x2s u0, u1;
                u0.x = acc1[4 * q + 0] + bv.x; u0.y = acc1[4 * q + 1] + bv.y;
                u1.x = acc1[4 * q + 2] + bv.z; u1.y = acc1[4 * q + 3] + bv.w;
                u0 = gelu2s(u0);
                u1 = gelu2s(u1);
                acc1[4 * q + 0] = u0.x; acc1[4 * q + 1] = u0.y; acc1[4 * q + 2] = u1.x; acc1[4 * q + 3] = u1.y;
            }
            // ---- Out^T += W2[:, tile t] . H^T tile: two k16 steps ---------------------------------------------------------
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r0 = 8 * s;
                unsigned xh[2], yh[2], xl[2], yl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    split_pair(acc1[r0 + 2 * e], acc1[r0 + 2 * e + 1], xh[e], xl[e]);
                    split_pair(acc1[r0 + 4 + 2 * e], acc1[r0 + 4 + 2 * e + 1], yh[e], yl[e]);
                }
                unsigned fhh[4], fll[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto rh = __builtin_amdgcn_permlane32_swap(xh[e], yh[e], false, false);
                    fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                    if (SPLIT == 3) {
                        auto rl = __builtin_amdgcn_permlane32_swap(xl[e], yl[e], false, false);
                        fll[e] = rl[0]; fll[2 + e] = rl[1];
                    }
                }
                const bf16x8 ph = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                bf16x8 pl;
                if (SPLIT == 3) pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fll));
#pragma unroll
                for (int ot = 0; ot < NOT; ++ot) {
                    const int cur = (s * NOT + ot) & 1, nxt = cur ^ 1;
                    if (ot + 1 < NOT || s == 0) {
                        const char* nx = ot + 1 < NOT ? w2_addr(ot + 1, s) : w2_addr(0, 1);
                        w2f[nxt][0] = *reinterpret_cast<const bf16x8*>(nx);
                        if (SPLIT == 3) w2f[nxt][1] = *reinterpret_cast<const bf16x8*>(nx + W2LO);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (SPLIT == 3) {
                        acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[cur][1], ph, acc2[ot], 0, 0, 0);
                        acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[cur][0], pl, acc2[ot], 0, 0, 0);
                    }
                    acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[cur][0], ph, acc2[ot], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue: x[frame][c] += gamma[c] (out + b2[c]); channel(r) = 32 ot + (r & 3) + 8 (r >> 2) + 4 fh ----------
        if (m_cur < a.M) {
            float* xr = a.x + a.img.at(m_cur);
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot) {
                float4 xo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) xo[q] = *reinterpret_cast<const float4*>(xr + 32 * ot + 8 * q + 4 * fh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * ot + 8 * q + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(vb2 + c0);
                    const float4 gv = *reinterpret_cast<const float4*>(vga + c0);
                    float4 o = xo[q];
                    o.x += gv.x * (acc2[ot][4 * q + 0] + bv.x);
                    o.y += gv.y * (acc2[ot][4 * q + 1] + bv.y);
                    o.z += gv.z * (acc2[ot][4 * q + 2] + bv.z);
                    o.w += gv.w * (acc2[ot][4 * q + 3] + bv.w);
                    *reinterpret_cast<float4*>(xr + c0) = o;
                }
            }
        }
        // the counted waits of the tile loop assume only DMA pieces are outstanding: drain this pass's loads / stores
        // (and with them the S-1 tiles already prefetched for the next pass, which have had a whole epilogue to land)
        wait_vmcnt<0>();
    }
}

template <int C, int SPLIT, int NW, int S, int TPB>
static hipError_t ffn_stream_go(const FfnStreamArgs& a, hipStream_t st) {
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
    constexpr size_t lds = (size_t)S * TPB * NARR * 128 * C + (size_t)7 * C * 4;
    static_assert(lds <= 160 * 1024, "ring exceeds LDS");
    auto kern = codec_ffn_stream_kernel<C, SPLIT, NW, S, TPB>;
    static bool done = false;
    static int cus = 256;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        done = true;
    }
    const int npass = (a.M + NW * 32 - 1) / (NW * 32);
    const int grid = npass < cus ? npass : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return hipGetLastError();
}

// C in {128, 256}; w1 [F][C] split bf16, w2t tile-major [F/32][C][32] split bf16 (launch_w2_tile_pack)
hipError_t launch_codec_ffn_stream(float* x, RowMap img, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo,
                                   const float* b1, const bf16_t* w2thi, const bf16_t* w2tlo, const float* b2, const float* gamma,
                                   int M, int C, int F, float eps, int split, hipStream_t st) {
    if (!(C == 128 || C == 256) || F != 4 * C || img.ld % 4 || img.off % 4 || (img.rpb && img.bstride % 4)) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    FfnStreamArgs a{x, img, norm_w, w1hi, w1lo, b1, w2thi, w2tlo, b2, gamma, M, eps};
    ProfScope ps(st, C == 128 ? "codec_ffn_stream<128>" : "codec_ffn_stream<256>", 4.0 * M * (double)C * F, 8.0 * M * C + 8.0 * (double)C * F);
    if (C == 128) return split == 3 ? ffn_stream_go<128, 3, 8, 4, 1>(a, st) : ffn_stream_go<128, 1, 8, 4, 1>(a, st);
    return split == 3 ? ffn_stream_go<256, 3, 4, 2, 1>(a, st) : ffn_stream_go<256, 1, 4, 4, 1>(a, st);
}

// out[(t * C + c) * 32 + k] = in[c * F + 32 t + k]   (W2 [C][F] -> hidden-tile-major)
__global__ void w2_tile_pack_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int C, int F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * F) return;
    const int k = (int)(i % 32);
    const long r = i / 32;
    const int c = (int)(r % C), t = (int)(r / C);
    out[i] = in[(long)c * F + 32 * t + k];
}
hipError_t launch_w2_tile_pack(const bf16_t* in, bf16_t* out, int C, int F, hipStream_t st) {
    const long n = (long)C * F;
    hipLaunchKernelGGL(w2_tile_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, C, F);
    return hipGetLastError();
}
