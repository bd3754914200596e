// LAB: copy of ../codec_ffn_stream.hip (make LAB=1 compiles THIS file instead): every timing-experiment branch (FS_ELIM_*, FS_LIN_STORE,
// LAB: FS_NOBARRIER, ... — several give wrong results by design) and the FS_TIMELINE stamps.  The shipped file is this one with those switches
// LAB: resolved as "not defined" (tools/strip_lab.py; tests/test_host_cpu.py checks that the two have not drifted apart).
// Fused codec FFN for the stages whose weights do NOT fit in LDS (C = 128, 256), gfx950:
//     x += ffn_gamma * ( W2 . gelu( W1 . RMSNorm(x; g, eps) + b1 ) + b2 )
//
// Same wave-level structure as codec_ffn_wave.hip — each wave owns 32 frames, both products are computed transposed
// (lane = frame), the normalised input and the 4C-wide hidden live in registers, v_permlane32_swap turns GELU'd
// accumulators into the B fragments of the second product — but the weights stream through an LDS ring:
//   slot(i) = [ W1 rows 32i..32i+32 (32 x C) | W2 columns of hidden tile i-2 (C x 32, from a tile-major repack) ], hi and lo,
// i.e. exactly what pipeline step i needs (first product of tile i, second product of tile i-2; see the kernel).  All
// waves of the workgroup consume slot i for their own frames, so the ring is joined by one s_barrier per step (gemm3's
// protocol: own DMA pieces landed -> barrier -> refill the slot everybody just left).  The workgroup is persistent (walks
// passes of NW x 32 frames) and the ring keeps running across passes.  Per pass the L2 -> LDS weight traffic is 32 C^2
// bytes for NW x 32 frames, and nothing but x itself touches HBM (the unfused C = 256 path moved the 4C-wide hidden
// through HBM twice).
#include "gemm3.hpp"
#include "kernels.hpp"
#include "prof.hpp"
#include <type_traits>

extern int g_persist_mask;   // engine.hip: which persistent kernels the throughput-mode grid cap applies to (1 streamed FFN, 2 one-pass / wave FFN, 4 upsample)

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

// ---------------------------------------------------------------------------------------------------------------------
// Software pipeline: ring step i of a pass runs  P1(i) || GELU(i-1) || P2(i-2)  (first product of hidden tile i,
// activation + bf16 split of tile i-1, second product of tile i-2), i = 0 .. NT1+1.  The three are independent, so the
// whole GELU — written with plain (non-packed) fp32 instructions, which unlike v_pk_* run in the shadow of MFMAs
// (tools/ubench/mfma_valu.hip) — is cut into 42 small tasks that are dealt out one after each MFMA.  Ring slot i holds
// {W1 tile i, W2 tile i-2}; one barrier per step as before.  Weight fragments are read two MFMA groups ahead.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cvt_pk_bf16s(float a, float b) { return cvt_pk_bf16(a, b); }

#ifdef FS_TIMELINE   // debug build (tools/ffn_stream_timeline.py): wave 0 of every workgroup stamps the shader clock through its first FS_TL_PASSES
                     // passes — into LDS (a global store per stamp would join the counted vmcnt waits of the ring), dumped when the kernel ends
#define FS_TL_PASSES 3
#define FS_TL_N 160
static __device__ unsigned long long fs_tl_buf[256 * FS_TL_PASSES * FS_TL_N];
#define FS_STAMP(i) do { if (tid == 0 && p < FS_TL_PASSES) tl[p * FS_TL_N + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FS_STAMPR(i) do { if (tid == 0 && p < FS_TL_PASSES) tl[p * FS_TL_N + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int smtts_debug_read_fs_timeline(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs_tl_buf), (size_t)n * 8);
}
extern "C" int smtts_debug_clear_fs_timeline(void) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(fs_tl_buf)) != hipSuccess) return 1;
    return (int)hipMemset(q, 0, sizeof(unsigned long long) * 256 * FS_TL_PASSES * FS_TL_N);
}
#else
#define FS_STAMP(i) do { } while (0)
#define FS_STAMPR(i) do { } while (0)
#endif

// (Round 4 measured a variant that takes the first product's operand from the mixer — fp16 rows by LDS-DMA into a wave-private tile, one
// pass ahead, no fp32 tile at the top of the pass: FFN 282 -> 268 us (C = 128) and 242 -> 222 us (C = 256), but the mixer that has to
// write those rows 121 -> 163 and 73 -> 93 us — a net loss, and the per-pass timeline shows why the FFN gains so little: the wait only
// moves to the residual re-read, which was an L2 hit behind the top-of-pass load and is a cold HBM read without it.  Commit aefe457,
// profiles/r04k_*; the instrumented / elimination builds live in exp/codec_ffn_stream_lab.hip: make LAB=1, tools/ffn_stream_timeline.py.)
template <int C, int SPLIT, int NW, int S>
__global__ __launch_bounds__(NW * 64) void codec_ffn_stream_kernel(FfnStreamArgs a) {
    constexpr int F = 4 * C;
    constexpr int KK1 = C / 16;             // k16 steps of the first product
    constexpr int NT1 = F / 32;             // hidden tiles
    constexpr int NOT = C / 32;             // output (channel) tiles
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
    constexpr int NPASS = SPLIT == 3 ? 3 : 1;
#ifndef FFN_GELU3
#define FFN_GELU3 1
#endif
    constexpr bool G3 = FFN_GELU3 && SPLIT != 3;  // three-term erfc (A&S 7.1.25) where the hidden is rounded to 16 bits anyway
    constexpr int RB1 = 2 * C;              // bytes per W1 row (256 / 512)
    constexpr int CPR1 = RB1 / 16;          // 16-B chunks per W1 row (16 / 32)
    constexpr int W1T = 32 * RB1;           // bytes of one W1 tile image per array (= 64 C)
    constexpr int W2T = C * 64;             // bytes of one W2 tile image per array (C rows x 64 B)
    constexpr int SLOT = NARR * (W1T + W2T);
    constexpr int NSTEP = NT1 + 2;          // ring steps per pass
    constexpr int PIECES = SLOT / 1024;     // DMA pieces per slot
    constexpr int PW = PIECES / NW;         // per wave
    constexpr int HALF = NARR * (W1T / 1024);  // pieces [0, HALF) are W1, [HALF, PIECES) W2
    constexpr int OFF_V = S * SLOT;         // b1[F] b2[C] gamma[C] norm_w[C] (fp32)

    static_assert(PW * NW == PIECES, "DMA pieces must divide over the waves");
    static_assert((S - 2) * PW <= 63 && S >= 2, "vmcnt immediate");
    static_assert(NT1 % 2 == 0 && NT1 >= 4, "pipeline assumes an even number of hidden tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)SM_LPTR(smem);

    float* vb1 = reinterpret_cast<float*>(smem + OFF_V);
    float* vb2 = vb1 + F;
    float* vga = vb2 + C;
    float* vnw = vga + C;
#ifdef FS_TIMELINE
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(vnw + C);
    for (int i = tid; i < FS_TL_PASSES * FS_TL_N; i += NW * 64) tl[i] = 0;
#endif
    for (int i = tid; i < F; i += NW * 64) vb1[i] = a.b1[i];
    for (int i = tid; i < C; i += NW * 64) { vb2[i] = a.b2[i]; vga[i] = a.gamma[i]; vnw[i] = a.norm_w[i]; }

    // ---- this wave's DMA pieces: source pointer at hidden tile 0 and LDS offset inside a slot -------------------
    unsigned soff[PW];  // per-lane BYTE offset inside the array's tile (the array base + tile offset are scalar)
    unsigned dst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave * PW + i;  // wave-uniform piece index inside the slot
        if (q < HALF) {               // W1 tile: rows of RB1 bytes, 1024 / RB1 rows per piece
            const int arr = q / (W1T / 1024), j = q % (W1T / 1024);
            const int r = j * (1024 / RB1) + lane / CPR1, pos = lane % CPR1;
            const int c = (pos & ~15) | ((pos ^ r) & 15);
            soff[i] = (unsigned)(r * C + c * 8) * 2u;
            dst[i] = arr * W1T + j * 1024;
        } else {                      // W2 tile: 64-B rows, 16 rows per piece
            const int q2 = q - HALF;
            const int arr = q2 / (W2T / 1024), j = q2 % (W2T / 1024);
            const int r = j * 16 + (lane >> 2), pos = lane & 3;
            const int c = pos ^ ((r >> 2) & 3);
            soff[i] = (unsigned)(r * 32 + c * 8) * 2u;
            dst[i] = NARR * W1T + arr * W2T + j * 1024;
        }
    }
    auto dma16 = [&](const bf16_t* sbase, unsigned voff, unsigned lds_dst) {  // scalar base + 32-bit lane offset
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_dst)
            : "memory");
    };
    // ring position i -> step si = i % NSTEP of a pass: W1 tile si (clamped, unused in the last two steps), W2 tile si - 2
    auto issue = [&](int i) {
        const int si = i % NSTEP;
        const int t1 = si < NT1 ? si : NT1 - 1, t2 = si >= 2 ? si - 2 : 0;
        const unsigned st = lds0 + (unsigned)((i % S) * SLOT);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int q = wave * PW + p;  // wave-uniform
            const bool is_w2 = q >= HALF;
            const int arr = is_w2 ? (q - HALF) / (W2T / 1024) : q / (W1T / 1024);
            const bf16_t* base = is_w2 ? (arr ? a.w2tlo : a.w2thi) : (arr ? a.w1lo : a.w1hi);
            dma16(base + (long)(is_w2 ? t2 : t1) * 32 * C, soff[p], st + (unsigned)__builtin_amdgcn_readfirstlane((int)dst[p]));
        }
    };

#ifdef FS_PHASE_TICKS   // (A/B builds: with two 4-wave workgroups per CU, hold the second half of the grid back by this many 10-ns ticks so
                        // that the two workgroups of a CU do their tile I/O at different times)
    if (blockIdx.x >= gridDim.x / 2) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(FS_PHASE_TICKS)) __builtin_amdgcn_s_sleep(32);
    }
#endif
    const int npass_total = (a.M + NW * 32 - 1) / (NW * 32);
    const int my_passes = blockIdx.x < npass_total ? (npass_total - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int total = my_passes * NSTEP;  // ring slots this workgroup will consume
#pragma unroll 1
    for (int s = 0; s < S - 1; ++s)
        if (s < total) issue(s);

    // fragment byte offsets inside a slot
    // W1 fragment of k16 step kk: row fr, 16-B chunk c = 2 kk + fh stored at (c & ~15) | ((c ^ fr) & 15).  (c ^ fr) & 15 =
    // (2 kk & 15) ^ ((fh ^ fr) & 15), and fr * RB1 has no bits below 256, so the byte offset is (w1_a0 ^ ((2 kk & 15) << 4)) +
    // (2 kk >> 4) * 256: one v_xor with a constant per read instead of KK1 address registers
    const int w1_a0 = fr * RB1 + (((fh ^ fr) & 15) << 4);
    int w2_off[2];  // (s, ot): w2_off[s] + ot * 32 * 64  (the swizzle term (row >> 2) & 3 is the same for row and row + 32)
#pragma unroll
    for (int s = 0; s < 2; ++s) w2_off[s] = NARR * W1T + fr * 64 + (((2 * s + fh) ^ ((fr >> 2) & 3)) << 4);

    int it = 0;  // ring position
    float4 xa[KK1][2];
    auto load_x = [&](int p) {   // this lane's frame of pass p: channels 16 kk + 8 fh + (0..7)
        const int mm = ((blockIdx.x + p * (int)gridDim.x) * NW + wave) * 32 + fr;
        const float* xr = a.x + a.img.at(mm < a.M ? mm : a.M - 1);
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
#ifdef FS_ELIM_XIN   // (timing experiment: no global read of the tile)
            const float f0 = (float)(lane + kk) * 0.01f + (float)(size_t)xr * 1e-30f;
            xa[kk][0] = make_float4(f0, f0 + 1.f, f0 - 1.f, f0 * 0.5f);
            xa[kk][1] = make_float4(-f0, f0 + 2.f, f0 - 2.f, f0 * 0.25f);
#else
            xa[kk][0] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh);
            xa[kk][1] = *reinterpret_cast<const float4*>(xr + 16 * kk + 8 * fh + 4);
#endif
        }
    };
    // x I/O of a pass used to cost three exposed memory round trips (tile read at the top; residual re-read, then the drain of
    // its stores at the bottom — elimination builds, profiles/r02x_*: 120 of 313 us at C = 128).  Two of them are hidden now:
    //  FS_XNEXT     the NEXT pass's tile is requested right behind this pass's stores (the accumulators and the re-read tile
    //               are dead by then, so it costs no registers) and arrives under the one drain both need anyway;
    //  FS_XO_EARLY  the residual re-read is issued two ring steps before the end of the pass, into the registers the
    //               normalised input fragments and the first-product accumulators have just left; the counted waits of those
    //               two steps allow for the XL younger loads.
    // Measured (profiles/r03u_*, us per launch, old / early re-read / next-tile request / both): C = 128: 304.6 / 290.3 / 302.8 /
    // 300.4; C = 256: 238.8 / 237.0 / 232.0 / 238.1 — far less than the 120 us the elimination builds promised: what a pass
    // waits for is not the latency of its own requests but the BURST — every workgroup of the launch reaches its write-back
    // within the same few microseconds (32 MB of stores, then 32 MB of reads, per round of passes), and nothing computes while
    // HBM serves it.  Each width keeps the variant that helps it (the lab copy, exp/codec_ffn_stream_lab.hip, takes -DFS_XNEXT=0/1 -DFS_XO_EARLY=0/1 for A/B builds).
#ifdef FS_XNEXT
    constexpr bool XNEXT = FS_XNEXT;
#else
    constexpr bool XNEXT = C == 256;
#endif
#ifdef FS_XO_EARLY
    constexpr bool XO_EARLY = FS_XO_EARLY;
#else
    constexpr bool XO_EARLY = C == 128;
#endif
    constexpr int XL = XO_EARLY ? 4 * NOT : 0;   // loads of the early re-read (per lane)
    static_assert((S - 2) * PW + XL <= 63, "vmcnt immediate");
    if (XNEXT && my_passes > 0) load_x(0);
#pragma unroll 1
    for (int p = 0; p < my_passes; ++p) {
        const int pass = blockIdx.x + p * gridDim.x;
        const int m_cur = (pass * NW + wave) * 32 + fr;
        FS_STAMP(0);
        FS_STAMPR(150);
        // ---- x -> RMSNorm -> split bf16 B fragments (lane = frame, channels 16 kk + 8 fh + e) -------------------
        bf16x8 nh[KK1], nl[KK1];
        {
            if (!XNEXT) load_x(p);
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
                for (int e = 0; e < 4; ++e) {
                    if (SPLIT == PREC_F16) { nhp[e] = cvt_pk_f16_raw(v[2 * e], v[2 * e + 1]); nlp[e] = 0; }   // |n| <= sqrt(C) max |g|: certified at finalize
                    else split_pair(v[2 * e], v[2 * e + 1], nhp[e], nlp[e]);
                }
                nh[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nhp));
                nl[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(nlp));
            }
        }
        floatx16 acc2[NOT];
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;
#ifdef FS_TIMELINE
        asm volatile("s_nop 0" :: "v"(nh[KK1 - 1]));   // (the stamp stays behind the tile load + norm)
#endif
        FS_STAMP(1);   // tile loaded, normalised, converted

        struct Frags { bf16x8 h[2], l[2]; };  // B fragments (two k16 steps) of one activated hidden tile
        floatx16 H0, H1;
        Frags F0, F1;

        // one ring step; ti = hidden tile of the first product (step index), do_* select what the step contains
        auto step = [&](auto do_p1, auto do_g, auto do_p2, int ti, floatx16& hw, const floatx16& hr, Frags& fw, const Frags& fr_, auto younger) {
            constexpr bool P1 = decltype(do_p1)::value, G = decltype(do_g)::value, P2 = decltype(do_p2)::value;
            constexpr int YL = decltype(younger)::value;   // loads issued by this wave AFTER the DMA pieces of this step's slot (early re-read)
            // ---- ring hand-over (one barrier per step) ----
            if (it + S - 1 <= total)
                wait_vmcnt<(S - 2) * PW + YL>();
            else
                wait_vmcnt<0>();
            FS_STAMP(2 + 4 * ti);   // own DMA pieces of this step's slot landed
#ifndef FS_NOBARRIER   // (timing experiment only: without the barrier the ring hand-over is a data race)
            __builtin_amdgcn_s_barrier();  // everybody's pieces of slot `it` landed; everybody left slot it-1 -> it is free
#endif
            FS_STAMP(3 + 4 * ti);   // barrier passed
            if (it + S - 1 < total) issue(it + S - 1);
            FS_STAMP(4 + 4 * ti);   // next slot's DMAs issued
            const char* sl = smem + (it % S) * SLOT;
            ++it;

            constexpr int NG1 = P1 ? KK1 : 0, NG2 = P2 ? 2 * NOT : 0, NG = NG1 + NG2, NCH = NG * NPASS;
            // weight fragment of MFMA group g: first product k16 step g, or second product (s, ot) = ((g - NG1) / NOT, (g - NG1) % NOT)
#ifndef FS_PFD128
#define FS_PFD128 1
#endif
            constexpr int PFD = C >= 256 ? 3 : (SPLIT != 3 ? FS_PFD128 : 1);  // fragment reads run PFD MFMA groups ahead (PFD + 1 register buffers)
            constexpr int NB = PFD + 1;
            bf16x8 wf[NB][2];
            auto read_frag = [&](int g) {
                const char* ad;
                int lo;
                if (g < NG1) { ad = sl + (w1_a0 ^ (((2 * g) & 15) << 4)) + ((2 * g) >> 4) * 256; lo = W1T; }
                else { const int g2 = g - NG1; ad = sl + w2_off[g2 / NOT] + (g2 % NOT) * 32 * 64; lo = W2T; }
#ifdef FS_ELIM_FRAG   // (timing experiments only, results are wrong: FS_ELIM_*)
                wf[g % NB][0] = nh[g % KK1]; wf[g % NB][1] = nl[g % KK1]; (void)ad; (void)lo;
#else
                wf[g % NB][0] = *reinterpret_cast<const bf16x8*>(ad);
                if (SPLIT == 3) wf[g % NB][1] = *reinterpret_cast<const bf16x8*>(ad + lo);
#endif
            };
#pragma unroll
            for (int g = 0; g < PFD; ++g)
                if (g < NG) read_frag(g);
            if (P1) {  // accumulators start from b1: row r of tile ti is hidden unit 32 ti + (r & 3) + 8 (r >> 2) + 4 fh
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(vb1 + 32 * ti + 8 * q + 4 * fh);
                    hw[4 * q + 0] = bv.x; hw[4 * q + 1] = bv.y; hw[4 * q + 2] = bv.z; hw[4 * q + 3] = bv.w;
                }
            }
            // ---- GELU of tile ti-1 as 42 tasks: exact-erf A&S 7.1.26, gelu(x) = x/2 + |x/2| erf(|x| / sqrt 2) ----
            // two halves of 42 micro-tasks (<= 4 VALU instructions each, so a task fits the 32-cycle shadow of ONE MFMA next to its
            // issue): values 8 s2 .. 8 s2 + 7 (= k16 step s2 of the second product) run S1a/b x8, S2a/b x8, four pair splits in two
            // parts and the lane swap in two parts before the other half starts, so only 8 values' temporaries are live
            float tt[8], ee[8];   // tt: t, then the result (in place)
            unsigned hiP[4], loP[4];
            auto task = [&](int k0) {
                const int s2 = k0 / 42, k = k0 % 42;
                if (k < 16) {                 // S1: t = 1 / (1 + p |x| / sqrt 2) | e = exp2(-x^2 / 2 log2 e)
                    const int v = k >> 1;
                    const float x = hr[8 * s2 + v];
                    if (k & 1) ee[v] = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));
                    else tt[v] = fast_rcp(fmaf(fabsf(x), G3 ? Gelu3::P : 0.3275911f * 0.70710678118654752f, 1.0f));
                } else if (k < 32) {          // S2: erfc polynomial (Horner, 4 fma) | erf and the result (4)
                    const int v = (k - 16) >> 1;
                    if (!(k & 1)) {
                        const float t = tt[v];
                        if (G3) tt[v] = t * fmaf(t, fmaf(t, Gelu3::A3, Gelu3::A2), Gelu3::A1);
                        else tt[v] = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
                    } else {
                        const float u = fmaf(-tt[v], ee[v], 1.0f);
                        const float hx = 0.5f * hr[8 * s2 + v];
                        tt[v] = fmaf(fabsf(hx), u, hx);
                    }
                } else if (k < 40) {          // S3: bf16 split of the four value pairs: hi | residual
                    const int j = (k - 32) >> 1;
                    const float rx = tt[2 * j], ry = tt[2 * j + 1];
                    if (!(k & 1)) hiP[j] = SPLIT == PREC_F16 ? cvt_pk_f16_satpos(rx, ry) : cvt_pk_bf16s(rx, ry);
                    else if (SPLIT == 3) loP[j] = cvt_pk_bf16s(rx - __uint_as_float(hiP[j] << 16), ry - __uint_as_float(hiP[j] & 0xffff0000u));
                } else {                      // lane swap of this k16 step: hi | lo
                    // lane half 0 needs hidden 0..7 of the k16 step, half 1 hidden 8..15: swap upper half of X with lower half of Y
                    if (k == 40) {
                        unsigned fhh[4];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            auto rh = __builtin_amdgcn_permlane32_swap(hiP[e], hiP[2 + e], false, false);
                            fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                        }
                        fw.h[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                    } else if (SPLIT == 3) {
                        unsigned fll[4];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            auto rl = __builtin_amdgcn_permlane32_swap(loP[e], loP[2 + e], false, false);
                            fll[e] = rl[0]; fll[2 + e] = rl[1];
                        }
                        fw.l[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fll));
                    }
                }
            };
            // ---- GELU for the single-array formats: GeluQ5 (common.hpp) — max(x, 0) - |x| 2^q(|x|), q of degree 5: 8 plain VALU
            // instructions and ONE transcendental per value (A&S: 13 / 11 and two).  With a third of the MFMAs these kernels are bound
            // by what their waves issue between MFMAs (SQ counters: profiles/r02d_pmc_codec_ffn_f16.txt), and packing value pairs
            // into v_pk_* did not help (profiles/r02i_ab_packed_gelu.txt: a packed fp32 instruction costs two plain ones here).
            // 29 tasks per half: Horner in two parts x8, exp2 + result x8 ... convert + saturate x4 pairs, lane swap.
            auto task_q5 = [&](int k0) {
                const int s2 = k0 / 29, k = k0 % 29;
                if (k < 16) {
                    const int v = k >> 1;
                    const float ax = fabsf(hr[8 * s2 + v]);
                    if (!(k & 1)) tt[v] = fmaf(fmaf(fmaf(GeluQ5::Q5, ax, GeluQ5::Q4), ax, GeluQ5::Q3), ax, GeluQ5::Q2);
                    else ee[v] = __builtin_amdgcn_exp2f(fmaf(fmaf(tt[v], ax, GeluQ5::Q1), ax, GeluQ5::Q0));
                } else if (k < 24) {
                    const int v = k - 16;
                    const float x = hr[8 * s2 + v];
                    tt[v] = fmaf(-fabsf(x), ee[v], relu_f(x));
                } else if (k < 28) {
                    const int j = k - 24;
                    hiP[j] = SPLIT == PREC_F16 ? cvt_pk_f16_satpos(tt[2 * j], tt[2 * j + 1]) : cvt_pk_bf16s(tt[2 * j], tt[2 * j + 1]);
                } else {
                    unsigned fhh[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        auto rh = __builtin_amdgcn_permlane32_swap(hiP[e], hiP[2 + e], false, false);
                        fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                    }
                    fw.h[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                }
            };
            // ---- the same formula on value PAIRS in packed fp16 (round 4, common.hpp gelu_q5_pk_*): v_pk_fma_f16 runs at the plain-VALU
            // rate in the MFMA shadow (tools/ubench/mfma_valu.hip KIND 6), 5 plain + 1 transcendental instruction per value
            // instead of 9 + 1.  17 tasks per half: per pair {convert, |.|, two Horner steps} {three Horner steps} {exp2 x 2, pack}
            // {max, fma -> the operand pair}, then the lane swap.  No clamp: the hidden's range is certified from the weights.
            unsigned hpP[4], axP[4], qP[4];
            auto task_pk = [&](int k0) {
                const int s2 = k0 / 17, k = k0 % 17;
                if (k < 16) {
                    const int j = k >> 2, ph = k & 3;
                    if (ph == 0) {
                        f32x2_t v;
                        v.x = hr[8 * s2 + 2 * j]; v.y = hr[8 * s2 + 2 * j + 1];
                        hpP[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
                        axP[j] = hpP[j] & 0x7fff7fffu;
                        const half2_t ax = __builtin_bit_cast(half2_t, axP[j]);
#if GELU_PK_DEG == 5
                        half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ5::Q5), ax, h2_splat(GeluQ5::Q4));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q3));
#elif GELU_PK_DEG == 4  // the default (common.hpp: degree-5 accuracy in the packed evaluation, one instruction fewer)
                        half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ4::Q4), ax, h2_splat(GeluQ4::Q3));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ4::Q2));
#else
                        half2_t q = __builtin_elementwise_fma(h2_splat(GeluQ3::Q3), ax, h2_splat(GeluQ3::Q2));
#endif
                        qP[j] = __builtin_bit_cast(unsigned, q);
                    } else if (ph == 1) {
                        const half2_t ax = __builtin_bit_cast(half2_t, axP[j]);
#if GELU_PK_DEG == 5
                        half2_t q = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, qP[j]), ax, h2_splat(GeluQ5::Q2));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q1));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ5::Q0));
#elif GELU_PK_DEG == 4
                        half2_t q = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, qP[j]), ax, h2_splat(GeluQ4::Q1));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ4::Q0));
#else
                        half2_t q = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, qP[j]), ax, h2_splat(GeluQ3::Q1));
                        q = __builtin_elementwise_fma(q, ax, h2_splat(GeluQ3::Q0));
#endif
                        qP[j] = __builtin_bit_cast(unsigned, q);
                    } else if (ph == 2) {
                        qP[j] = exp2_pk_f16(qP[j]);
                    } else {
                        hiP[j] = gelu_q5_pk_back(hpP[j], axP[j], qP[j]);
                    }
                } else {
                    unsigned fhh[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        auto rh = __builtin_amdgcn_permlane32_swap(hiP[e], hiP[2 + e], false, false);
                        fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                    }
                    fw.h[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                }
            };
#ifndef FFN_GELUQ5
#define FFN_GELUQ5 1
#endif
#ifndef FFN_GELU_PK16
#define FFN_GELU_PK16 1
#endif
            constexpr bool PK = FFN_GELUQ5 && G3;
            constexpr bool PK16 = FFN_GELU_PK16 && PK && SPLIT == PREC_F16;
            constexpr int NTASK = PK16 ? 34 : PK ? 58 : 84;
            if (NG == 0) {
                // (no such step: the first step has P1, the last has P2)
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + PFD < NG) read_frag(g + PFD);
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = g * NPASS + ps;
                    __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ELIM_MFMA
                    asm volatile("" :: "v"(wf[g % NB][0]));
                    if (g < NG1) { hw[g & 15] += 1.0f; } else { acc2[(g - NG1) % NOT][g & 15] += fr_.h[(g - NG1) / NOT][0] == (bf16_t)0.5f ? 1.f : 0.f; }
                    if (false) {
#else
                    if (g < NG1) {
#endif
                        // pass order: the two cross terms first, hi . hi last
                        if (SPLIT == 3 && ps == 0) hw = mfma16<SPLIT>(wf[g % NB][1], nh[g < NG1 ? g : 0], hw);
                        else if (SPLIT == 3 && ps == 1) hw = mfma16<SPLIT>(wf[g % NB][0], nl[g < NG1 ? g : 0], hw);
                        else hw = mfma16<SPLIT>(wf[g % NB][0], nh[g < NG1 ? g : 0], hw);
                    } else {
                        const int g2 = g - NG1, s2 = g2 / NOT, ot = g2 % NOT;
                        if (SPLIT == 3 && ps == 0) acc2[ot] = mfma16<SPLIT>(wf[g % NB][1], fr_.h[s2], acc2[ot]);
                        else if (SPLIT == 3 && ps == 1) acc2[ot] = mfma16<SPLIT>(wf[g % NB][0], fr_.l[s2], acc2[ot]);
                        else acc2[ot] = mfma16<SPLIT>(wf[g % NB][0], fr_.h[s2], acc2[ot]);
                    }
#ifdef FS_ELIM_GELU
                    if (G && c < 2) {   // bare conversion of the tile (keeps the first product alive): 8 cvt + the lane swap
                        const int s2 = c;
                        unsigned hp[4], fhh[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) hp[j] = cvt_pk_f16_satpos(hr[8 * s2 + 2 * j], hr[8 * s2 + 2 * j + 1]);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            auto rh = __builtin_amdgcn_permlane32_swap(hp[e], hp[2 + e], false, false);
                            fhh[e] = rh[0]; fhh[2 + e] = rh[1];
                        }
                        fw.h[s2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fhh));
                    }
#else
                    if (G)
#pragma unroll
                        for (int k = NTASK * c / NCH; k < NTASK * (c + 1) / NCH; ++k) {
                            if constexpr (PK16) task_pk(k); else if constexpr (PK) task_q5(k); else task(k);
                        }
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef FS_TIMELINE
            asm volatile("s_nop 0" :: "v"(acc2[0][0]), "v"(hw[0]));   // (the stamp stays behind the step's MFMAs)
#endif
            FS_STAMP(5 + 4 * ti);   // fragment reads + MFMAs + GELU tasks of the step issued
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        // step i: P1 -> H[i & 1]; GELU: H[(i+1) & 1] -> Fr[(i+1) & 1]; P2 reads Fr[i & 1]
        using Y0 = std::integral_constant<int, 0>;
        step(T_{}, F_{}, F_{}, 0, H0, H0, F1, F0, Y0{});  // (no GELU / second product yet: hr, fr_ unused)
        step(T_{}, T_{}, F_{}, 1, H1, H0, F0, F1, Y0{});
#pragma unroll 1
        for (int i = 2; i < NT1; i += 2) {
            step(T_{}, T_{}, T_{}, i, H0, H1, F1, F0, Y0{});
            step(T_{}, T_{}, T_{}, i + 1, H1, H0, F0, F1, Y0{});
        }
        // the residual re-read, in the accumulator layout (clamped row: every lane issues the same number of loads)
        float* const xr = a.x + a.img.at(m_cur < a.M ? m_cur : a.M - 1);
        float4 xo[NOT][4];
        auto load_xo = [&]() {
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
                for (int q = 0; q < 4; ++q) xo[ot][q] = *reinterpret_cast<const float4*>(xr + 32 * ot + 8 * q + 4 * fh);
        };
        if (XO_EARLY) load_xo();
        // step NT1's slot and the two behind it were issued before the re-read; with a two-slot ring step NT1 + 1's slot is issued
        // inside step NT1, i.e. AFTER it, and must be waited for without the allowance
        step(F_{}, T_{}, T_{}, NT1, H0, H1, F1, F0, std::integral_constant<int, XL>{});
        step(F_{}, F_{}, T_{}, NT1 + 1, H1, H0, F0, F1, std::integral_constant<int, (S > 2 ? XL : 0)>{});

        // ---- epilogue: x[frame][c] += gamma[c] (out + b2[c]); channel(r) = 32 ot + (r & 3) + 8 (r >> 2) + 4 fh ----------
        FS_STAMP(140);   // ring steps of the pass done
        if (!XO_EARLY) load_xo();   // every re-read of the tile is issued before the first store: one round trip, not one per channel tile
#ifndef FS_EPI_NOFENCE
        __builtin_amdgcn_sched_barrier(0);   // (keeps hipcc from sinking the loads back next to their stores)
#endif
#ifdef FS_ELIM_XOUT   // (timing experiment: the residual read-modify-write never happens, but the compiler cannot know)
        if (m_cur < a.M && a.eps < 0.f) {
#else
        if (m_cur < a.M) {
#endif
#pragma unroll
            for (int ot = 0; ot < NOT; ++ot) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * ot + 8 * q + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(vb2 + c0);
                    const float4 gv = *reinterpret_cast<const float4*>(vga + c0);
                    float4 o = xo[ot][q];
                    o.x += gv.x * (acc2[ot][4 * q + 0] + bv.x);
                    o.y += gv.y * (acc2[ot][4 * q + 1] + bv.y);
                    o.z += gv.z * (acc2[ot][4 * q + 2] + bv.z);
                    o.w += gv.w * (acc2[ot][4 * q + 3] + bv.w);
#ifdef FS_LIN_STORE   // (timing experiment, results wrong: the same bytes written as whole 1-KiB wave stores instead of 32-B pieces of 32 rows)
                    *reinterpret_cast<float4*>(a.x + a.img.at(0) + ((((long)(pass * NW + wave) * NOT + ot) * 4 + q) * 64 + lane) * 4) = o;
#else
                    *reinterpret_cast<float4*>(xr + c0) = o;
#endif
                }
            }
        }
        // the counted waits of the step loop assume only DMA pieces are outstanding (stores complete out of order with loads):
        // drain this pass's stores — and with them the next pass's tile, requested behind them
        // (tried in round 2: requesting it BEFORE this write-back and leaving the stores in flight — the 64 extra live
        // registers spill, 300 -> 321 us, profiles/r02x_*)
        __builtin_amdgcn_sched_barrier(0);   // (the request stays BEHIND the stores: hoisted above them it would be live next to the accumulators)
        // (unconditional — behind the last pass it re-reads a clamped row: a conditional request would keep the OLD tile's 64-128
        // registers live through the whole pass on the not-taken path, and spills)
        FS_STAMP(141);   // residual added, stores issued
        if (XNEXT) load_x(p + 1);
        wait_vmcnt<0>();
        FS_STAMP(142);   // stores (and the next tile's request) drained
        FS_STAMPR(151);
    }
#ifdef FS_TIMELINE
    __syncthreads();
    if (blockIdx.x < 256)
        for (int i = tid; i < FS_TL_PASSES * FS_TL_N; i += NW * 64) fs_tl_buf[blockIdx.x * FS_TL_PASSES * FS_TL_N + i] = tl[i];
#endif
}

template <int C, int SPLIT, int NW, int S>
static hipError_t ffn_stream_go(const FfnStreamArgs& a, hipStream_t st) {
    constexpr int NARR = SPLIT == 3 ? 2 : 1;
#ifdef FS_TIMELINE
    constexpr size_t lds = (size_t)S * NARR * 128 * C + (size_t)7 * C * 4 + FS_TL_PASSES * FS_TL_N * 8;
#else
    constexpr size_t lds = (size_t)S * NARR * 128 * C + (size_t)7 * C * 4;
#endif
    static_assert(lds <= 160 * 1024, "ring exceeds LDS");
    auto kern = codec_ffn_stream_kernel<C, SPLIT, NW, S>;
    static DevOnce once;
    int cus = 256;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }, &cus);
    if (e != hipSuccess) return e;
    if (!(g_persist_mask & 1)) cus = once.real_cus();   // (A/B: which persistent kernels the throughput-mode grid cap applies to)
    const int npass = (a.M + NW * 32 - 1) / (NW * 32);
    // NW = 4 (one wave per SIMD per workgroup): two workgroups share a CU when the ring is small enough — their steps drift
    // freely against each other, only the four waves of one ring meet at its barrier
    const int per_cu = (NW == 4 && 2 * lds <= 160 * 1024 && C == 128) ? 2 : 1;
    const int grid = npass < cus * per_cu ? npass : cus * per_cu;
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
#ifndef FS_S128
#define FS_S128 4
#endif
#ifndef FS_NW128
#define FS_NW128 8
#endif
#ifndef FS_S256
#define FS_S256 4
#endif
    if (C == 128) return split == 3 ? ffn_stream_go<128, 3, 8, 4>(a, st) : split == PREC_F16 ? ffn_stream_go<128, 2, FS_NW128, FS_S128>(a, st) : ffn_stream_go<128, 1, 8, 4>(a, st);
    return split == 3 ? ffn_stream_go<256, 3, 4, 2>(a, st) : split == PREC_F16 ? ffn_stream_go<256, 2, 4, FS_S256>(a, st) : ffn_stream_go<256, 1, 4, 4>(a, st);
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
