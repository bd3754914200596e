// Joint attention on the matrix cores (split-bf16, fp32-class), gfx950.
//
// One workgroup (4 waves) = 32 queries of one (batch, head); keys/values stream through LDS in 64-key chunks
// with an online softmax, so Ktot is unbounded.  Inputs are the pre-normalised / pre-rotated q, k (qk_prep_kernel)
// and v rows of the packed projection buffer plus the cross-KV cache.  PREP = true folds qk_prep into the staging passes: the
// raw projection rows are RMS-normalised per head, scaled by the q_norm / k_norm weights and rotated (dit.py:95-108) on their
// way into LDS — a row's dims sit in one aligned group of DHP / 4 lanes and the rotation pairs (2i, 2i + 1) in one lane — so the
// projection buffer is read once and never rewritten (one launch and one fp32 round trip less per block; the self keys are
// normalised once per 32-query tile, which is noise next to the staging itself).
//
//   S^T[key][query] = K . Q^T       A = K chunk  [64 keys][DHP dims]  (row-major, dims contiguous)
//                                   B = Q tile   [32 queries][DHP]    -> every lane owns ONE query column,
//                                   so max / sum over keys are in-lane (+ one exchange with lane^32).
//   O^T[dim][query] = V^T . P^T     A = V^T chunk [DHP dims][64 keys]  (transposed while staging)
//                                   B = P^T fragments built in registers from the S^T accumulators with
//                                   v_permlane32_swap (no LDS round trip for P).
// Every wave computes the full S^T chunk (cheap: 48 MFMAs) so the softmax state is wave-local, and owns one 32-dim
// tile of O^T.  All operands are split into bf16 hi/lo and each product takes 3 MFMAs (lo*hi + hi*lo + hi*hi).
// LDS images use 16-B chunks XOR-swizzled by row so ds_read_b128 fragment reads are conflict free.
#include <type_traits>

#include "kernels.hpp"
#include "prof.hpp"

#ifndef SMTTS_DBG
#define SMTTS_DBG 0   // r03 debug variants of the fused-prep kernel (tools/sessions/r03a.sh); 0 = product code
#endif

// sum over the aligned group of TPR (16 or 32) lanes that holds one row, left in every lane of the group: DPP inside a row of 16
// lanes (quad permutes, then the two mirrors), one ds_swizzle for the other row — no ds_bpermute round trips
template <int TPR>
__device__ __forceinline__ float row_group_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1 0 3 2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2 3 0 1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror: the other quad of the 8
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: the other 8 of the 16
    if (TPR == 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
    return v;
}

// RoPE factors of this lane's 4 dims for position `pos` (cos = 1, sin = 0 outside the rotated dims or when `on` is false),
// loaded ahead of the row reduction so the two round trips overlap
struct Rope4 { float4 c, s; };
__device__ __forceinline__ Rope4 rope4_load(const AttnArgs& a, int pos, int d, bool on) {
    Rope4 r{make_float4(1.f, 1.f, 1.f, 1.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (SMTTS_DBG != 2 && on && d < a.rot_dim) {   // rot_dim % 4 == 0 (checked by the launcher)
        r.c = *reinterpret_cast<const float4*>(a.rope_cos + (long)pos * a.rot_dim + d);
        r.s = *reinterpret_cast<const float4*>(a.rope_sin + (long)pos * a.rot_dim + d);
    }
    return r;
}

// RMSNorm_head(row) * w, then RoPE on the pairs (2i, 2i + 1), for this lane's 4 consecutive dims (v = 0 in pad dims)
template <int DH, int TPR>
__device__ __forceinline__ float4 prep_row4(float4 v, float4 w4, const Rope4& r, float eps) {
#if SMTTS_DBG == 4
    const float rstd = 1.0f + eps;
#else
    const float ss = row_group_sum<TPR>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
#if SMTTS_DBG == 10   // no transcendental instruction at all: bit-trick seed + 4 Newton steps on the FMA pipe
    const float xx = ss * (1.0f / (float)DH) + eps;
    float rstd = __builtin_bit_cast(float, 0x5f3759df - (__builtin_bit_cast(int, xx) >> 1));
#pragma unroll
    for (int it = 0; it < 4; ++it) rstd = rstd * (1.5f - 0.5f * xx * rstd * rstd);
#else
    float rstd = __builtin_amdgcn_rsqf(ss * (1.0f / (float)DH) + eps);
#endif
#if SMTTS_DBG == 8
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(rstd));
#elif SMTTS_DBG == 11
    asm volatile("s_nop 0" : "+v"(rstd));
#elif SMTTS_DBG == 12
    asm volatile("s_nop 3" : "+v"(rstd));
#endif
#endif
    const float4 y = make_float4(v.x * rstd * w4.x, v.y * rstd * w4.y, v.z * rstd * w4.z, v.w * rstd * w4.w);
    return make_float4(y.x * r.c.x - y.y * r.s.x, y.y * r.c.y + y.x * r.s.y, y.z * r.c.z - y.w * r.s.z, y.w * r.c.w + y.z * r.s.w);
}

// RES = 0: one workgroup per 32-query tile, keys / values streamed chunk by chunk (any Ktot).  RES = n: one workgroup per (batch,
// head); ALL keys / values (at most n chunks of 64) are staged — and, with PREP, normalised — ONCE and stay in LDS while the
// workgroup walks its query tiles.  For grids that are several rounds of the streaming form (the teacher's 3B-row CFG batches:
// 576 workgroups that each re-stage the same 120 keys) this removes two thirds of the staging, which is most of this kernel's time.
// Same arithmetic in the same order per (query, key) as RES = 0: the two forms are bit-identical (test_kernels_gpu.py).
// The resident form runs TWO 4-wave groups: all eight waves stage the keys / values, then each group takes every other query tile
// (own Q image, own softmax state), so a (batch, head) of three tiles costs one staging + two tile rounds instead of three.
template <int DH, bool PREP, int RES>
__global__ __launch_bounds__(RES ? 512 : 256) void attention_mfma_kernel(AttnArgs a) {
    constexpr int DHP = DH <= 64 ? 64 : 128;  // padded head dim (K of QK^T), zero filled
    constexpr int KC = 64, QT = 32;
    constexpr int QPITCH = DHP * 2;            // bytes per row of the Q / K images
    constexpr int CPR = QPITCH / 16;           // 16-B chunks per row (8 or 16)
    constexpr int Q_ARR = QT * QPITCH, K_ARR = KC * QPITCH, V_ARR = DHP * 128;  // Vt rows: 64 keys x 2 B = 128 B
    constexpr int NRES = RES ? RES : 1;        // resident chunk buffers
    constexpr int NG = RES ? 2 : 1, NT = 256 * NG;   // query-tile groups of 4 waves, threads
    constexpr int OFF_K0 = NG * 2 * Q_ARR, OFF_V0 = OFF_K0 + NRES * 2 * K_ARR;
    constexpr int NDT = DHP / 32;              // 32-dim tiles of O^T (2 or 4); wave w owns tile w (w < NDT)
    constexpr int KS1 = DHP / 16;              // k16 steps of S^T
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int gtid = tid & 255;                                          // thread inside its 4-wave group
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);           // group (0 unless RES)
    const int w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);       // wave inside the group: owns O^T tile w
    const int OFF_Q = grp * 2 * Q_ARR;                                   // the group's own Q image
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = a.N, R = a.k_ref ? a.R : 0, P = a.k_text ? a.P : 0, Ktot = N + R + P;
    const float sm_scale = 1.0f / sqrtf((float)DH);
    const int fr = lane & 31, fh = lane >> 5;

    auto swz = [](int row, int c) { return CPR == 16 ? (c ^ (row & 15)) : (c ^ ((row >> 1) & 7)); };
#if SMTTS_DBG == 1
    for (int i = tid; i < a.dbg_lds_bytes / 4; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0x7fc00000u;
    __syncthreads();
#endif

    // ---- stage Q (pre-scaled), split hi/lo: thread -> (query, 4 dims); every load of the tile is issued before the first use
    auto stage_q = [&](int q0) {
        constexpr int TPR = DHP / 4, NQP = QT * TPR / 256;   // lanes per row, passes (2 or 4)
        const int d = (gtid % TPR) * 4;
        float4 qv[NQP];
        Rope4 qr[NQP];
        float4 qw4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (SMTTS_DBG == 3) qw4 = make_float4(1.f, 1.f, 1.f, 1.f);
        else if (PREP && d < DH) qw4 = *reinterpret_cast<const float4*>(a.qw + h * DH + d);
#pragma unroll
        for (int j = 0; j < NQP; ++j) {
            const int r = gtid / TPR + j * (256 / TPR), n = q0 + r;
#if SMTTS_DBG == 14 || SMTTS_DBG == 15   // every load unconditional (clamped), consumed only after ONE full wait (+ 16 wait states at 14)
            {
                const int nc = n < N ? n : N - 1, dc = d < DH ? d : DH - 4;
                qv[j] = *reinterpret_cast<const float4*>(a.q + (long)b * a.bs + (long)nc * a.rs + h * DH + dc);
                const int dr = dc < a.rot_dim ? dc : 0;
                qr[j].c = *reinterpret_cast<const float4*>(a.rope_cos + (long)nc * a.rot_dim + dr);
                qr[j].s = *reinterpret_cast<const float4*>(a.rope_sin + (long)nc * a.rot_dim + dr);
            }
            continue;
#endif
            qv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < N && d < DH) qv[j] = *reinterpret_cast<const float4*>(a.q + (long)b * a.bs + (long)n * a.rs + h * DH + d);
            if (PREP) qr[j] = rope4_load(a, n, d, n < N);
        }
#if SMTTS_DBG == 14 || SMTTS_DBG == 15
#if SMTTS_DBG == 14
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
        for (int j = 0; j < NQP; ++j) {
            const int n = q0 + gtid / TPR + j * (256 / TPR);
            asm volatile("" : "+v"(qv[j].x), "+v"(qv[j].y), "+v"(qv[j].z), "+v"(qv[j].w));
            asm volatile("" : "+v"(qr[j].c.x), "+v"(qr[j].c.y), "+v"(qr[j].c.z), "+v"(qr[j].c.w));
            asm volatile("" : "+v"(qr[j].s.x), "+v"(qr[j].s.y), "+v"(qr[j].s.z), "+v"(qr[j].s.w));
            if (!(n < N && d < DH)) qv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(n < N && d < a.rot_dim)) { qr[j].c = make_float4(1.f, 1.f, 1.f, 1.f); qr[j].s = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
#endif
#pragma unroll
        for (int j = 0; j < NQP; ++j) {
            const int r = gtid / TPR + j * (256 / TPR);
            float4 v = qv[j];
            if (SMTTS_DBG == 7) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); }
            if (PREP && SMTTS_DBG != 6) v = prep_row4<DH, TPR>(v, qw4, qr[j], a.eps);
            const float f[4] = {v.x * sm_scale, v.y * sm_scale, v.z * sm_scale, v.w * sm_scale};
            bf16x4 hh, ll;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hh[e] = (bf16_t)f[e]; ll[e] = (bf16_t)(f[e] - (float)hh[e]); }
            const int off = r * QPITCH + (swz(r, d >> 3) << 4) + (d & 7) * 2;
            *reinterpret_cast<bf16x4*>(smem + OFF_Q + off) = hh;
            *reinterpret_cast<bf16x4*>(smem + OFF_Q + Q_ARR + off) = ll;
        }
    };

    float m_run = -INFINITY, l_run = 0.f;  // per query (lane & 31); both lane halves keep identical copies
    floatx16 oacc;

    // ---- stage K chunk [key][dim] and V^T chunk [dim][key] of keys c0 .. c0 + 63 into the buffers at OFF_K / OFF_V, split hi/lo.
    // All global loads of the chunk are issued before the first LDS store (one memory round trip per chunk, not one per item).
    // A thread owns 4 dims of ITEMS CONSECUTIVE keys, so its V^T output is one 16-B (8 keys) or 8-B (4 keys) piece per dim row.
    auto stage_kv = [&](int c0, int OFF_K, int OFF_V) {
        {
            constexpr int TPR = DHP / 4;                 // threads per key row (16 or 32)
            constexpr int ITEMS = KC * TPR / NT;         // keys per thread (2, 4 or 8)
            const int r0 = (tid / TPR) * ITEMS, d = (tid % TPR) * 4;
            float4 kq[ITEMS], vq[ITEMS];
            Rope4 kr[PREP ? ITEMS : 1];
            float4 kw4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (SMTTS_DBG == 3) kw4 = make_float4(1.f, 1.f, 1.f, 1.f);
            else if (PREP && d < DH) kw4 = *reinterpret_cast<const float4*>(a.kw + h * DH + d);
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                if (PREP) kr[it] = rope4_load(a, c0 + r0 + it, d, c0 + r0 + it < N);
                int gk = c0 + r0 + it;
                gk = gk < Ktot ? gk : Ktot - 1;          // clamp: always a readable row; masked out by vmask
                const int dc = d < DH ? d : DH - 4;      // clamp inside the row; pad dims are zeroed below
                const float* kp;
                const float* vp;
                if (gk < N) {
                    const long base = (long)b * a.bs + (long)gk * a.rs + h * DH + dc;
                    kp = a.k + base; vp = a.v + base;
                } else if (gk < N + R) {
                    const long base = (((long)b * a.H + h) * R + (gk - N)) * DH + dc;
                    kp = a.k_ref + base; vp = a.v_ref + base;
                } else {
                    const long base = (((long)b * a.H + h) * P + (gk - N - R)) * DH + dc;
                    kp = a.k_text + base; vp = a.v_text + base;
                }
                kq[it] = *reinterpret_cast<const float4*>(kp);
                vq[it] = *reinterpret_cast<const float4*>(vp);
            }
            float vt[4][ITEMS];  // [dim e][key]
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int r = r0 + it;
                const bool real = (c0 + r < Ktot) && d < DH;
                if (SMTTS_DBG == 7) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); }
                if (PREP && SMTTS_DBG != 5 && c0 < N) {   // self keys arrive raw; the cross-KV cache was normalised when it was built.  (c0 is uniform:
                    const bool self = c0 + r < N;   // chunks with no self key skip the pass; lanes never diverge around the DPP sum)
                    float4 raw = kq[it];
                    if (d >= DH) raw = make_float4(0.f, 0.f, 0.f, 0.f);   // (clamped column: not part of the row)
                    const float4 kn = prep_row4<DH, TPR>(raw, kw4, kr[it], a.eps);
                    if (self) kq[it] = kn;
                }
                const float kf[4] = {kq[it].x, kq[it].y, kq[it].z, kq[it].w};
                bf16x4 kh, kl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float kv = real ? kf[e] : 0.f;
                    kh[e] = (bf16_t)kv;
                    kl[e] = (bf16_t)(kv - (float)kh[e]);
                }
                const int koff = r * QPITCH + (swz(r, d >> 3) << 4) + (d & 7) * 2;
                *reinterpret_cast<bf16x4*>(smem + OFF_K + koff) = kh;
                *reinterpret_cast<bf16x4*>(smem + OFF_K + K_ARR + koff) = kl;
                vt[0][it] = real ? vq[it].x : 0.f; vt[1][it] = real ? vq[it].y : 0.f;
                vt[2][it] = real ? vq[it].z : 0.f; vt[3][it] = real ? vq[it].w : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // transpose: Vt[dim][key]; 128-B rows, 16-B chunk = key >> 3
                const int dr = d + e;
                const int voff = dr * 128 + (((r0 >> 3) ^ ((dr >> 1) & 7)) << 4) + (r0 & 7) * 2;
                typedef bf16_t bf16xI __attribute__((ext_vector_type(ITEMS)));
                bf16xI vh, vl;
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    vh[it] = (bf16_t)vt[e][it];
                    vl[it] = (bf16_t)(vt[e][it] - (float)vh[it]);
                }
                *reinterpret_cast<bf16xI*>(smem + OFF_V + voff) = vh;
                *reinterpret_cast<bf16xI*>(smem + OFF_V + V_ARR + voff) = vl;
            }
        }
    };
    // key validity of a chunk as a 64-bit mask (lane = key), identical in every wave
    auto chunk_mask = [&](int c0) -> unsigned long long {
        bool kval = false;
        const int gk = c0 + lane;
        if (gk < Ktot) {
            const uint8_t* mk = gk < N ? a.mask_self : (gk < N + R ? a.mask_ref : a.mask_text);
            const int mi = gk < N ? b * N + gk : (gk < N + R ? b * R + (gk - N) : b * P + (gk - N - R));
            kval = !mk || mk[mi];
        }
        return __ballot(kval);
    };

    // ---- one chunk of keys against the staged query tile: S^T, online softmax, O^T += V^T P^T ---------------------------------
    auto compute_chunk = [&](int OFF_K, int OFF_V, unsigned long long vmask) {
        // ---- S^T chunk: 2 key tiles x 32 queries ----------------------------------------------------------------
        floatx16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int c = ks * 2 + fh;
            const int qoff = OFF_Q + fr * QPITCH + (swz(fr, c) << 4);
            const bf16x8 qh = *reinterpret_cast<const bf16x8*>(smem + qoff);
            const bf16x8 ql = *reinterpret_cast<const bf16x8*>(smem + qoff + Q_ARR);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kr = t * 32 + fr;
                const int koff = OFF_K + kr * QPITCH + (swz(kr, c) << 4);
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(smem + koff);
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(smem + koff + K_ARR);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, s[t], 0, 0, 0);
            }
        }
        // ---- online softmax over keys for this lane's query: key(t, r) = 32 t + (r&3) + 8 (r>>2) + 4 fh ------------
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const bool ok = (vmask >> key) & 1ull;
                s[t][r] = ok ? s[t][r] : -INFINITY;
                cm = fmaxf(cm, s[t][r]);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float m_new = fmaxf(m_run, cm);
        float alpha = 1.f, csum = 0.f;
        const bool live = m_new != -INFINITY;
        if (live) alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (live && s[t][r] != -INFINITY) ? __expf(s[t][r] - m_new) : 0.f;
                s[t][r] = p;
                csum += p;
            }
        csum += __shfl_xor(csum, 32, 64);
        l_run = l_run * alpha + csum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;

        // ---- O^T tile w += V^T . P^T over the 64 keys (4 k16 steps) ---------------------------------------------
        if (w < NDT) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int t = ks >> 1, r0 = (ks & 1) * 8;
                unsigned xh[2], yh[2], xl[2], yl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float pa = s[t][r0 + 2 * e], pb = s[t][r0 + 2 * e + 1];
                    const float pc = s[t][r0 + 4 + 2 * e], pd = s[t][r0 + 4 + 2 * e + 1];
                    const bf16_t ah = (bf16_t)pa, bh = (bf16_t)pb, ch = (bf16_t)pc, dh_ = (bf16_t)pd;
                    const bf16_t al = (bf16_t)(pa - (float)ah), bl = (bf16_t)(pb - (float)bh);
                    const bf16_t cl = (bf16_t)(pc - (float)ch), dl = (bf16_t)(pd - (float)dh_);
                    auto pk = [](bf16_t lo, bf16_t hi) {
                        return (unsigned)__builtin_bit_cast(unsigned short, lo) |
                               ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                    };
                    xh[e] = pk(ah, bh); yh[e] = pk(ch, dh_);
                    xl[e] = pk(al, bl); yl[e] = pk(cl, dl);
                }
                // half 0 needs keys 0..7 of the step, half 1 keys 8..15: swap the upper half of X with the lower half of Y
                unsigned fh_[4], fl_[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto rh = __builtin_amdgcn_permlane32_swap(xh[e], yh[e], false, false);
                    auto rl = __builtin_amdgcn_permlane32_swap(xl[e], yl[e], false, false);
                    fh_[e] = rh[0]; fh_[2 + e] = rh[1];
                    fl_[e] = rl[0]; fl_[2 + e] = rl[1];
                }
                const bf16x8 ph = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fh_));
                const bf16x8 pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fl_));
                const int vr = w * 32 + fr;
                const int voff = OFF_V + vr * 128 + (((ks * 2 + fh) ^ ((vr >> 1) & 7)) << 4);
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(smem + voff);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(smem + voff + V_ARR);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc, 0, 0, 0);
            }
        }
    };

    // ---- normalise, gate, store: lane = query fr, rows = dims 32 w + (r&3) + 8 (r>>2) + 4 fh ---------------------------
    auto finish = [&](int q0) {
    const int n = q0 + fr;
    if (w < NDT && n < N) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        const long gbase = (long)b * a.bs + (long)n * a.rs + h * DH;
        const long obase = (long)b * a.obs + (long)n * a.ors + h * DH;
        // accumulator rows 4 q .. 4 q + 3 are four consecutive dims 32 w + 8 q + 4 fh + (0..3): one 16-B gate load and one
        // 8-B store per array for each group (DH % 4 == 0, so a group is entirely inside or outside the head)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d0 = 32 * w + 8 * q + 4 * fh;
            if (d0 < DH) {
                const float4 g4 = *reinterpret_cast<const float4*>(a.gate + gbase + d0);
                const float val[4] = {oacc[4 * q + 0] * inv * sigmoid_f(g4.x), oacc[4 * q + 1] * inv * sigmoid_f(g4.y),
                                      oacc[4 * q + 2] * inv * sigmoid_f(g4.z), oacc[4 * q + 3] * inv * sigmoid_f(g4.w)};
                if (a.out_hi) {
                    store_split4(a.out_hi, a.out_lo, obase + d0, make_float4(val[0], val[1], val[2], val[3]));
                } else {
                    *reinterpret_cast<float4*>(a.out + obase + d0) = make_float4(val[0], val[1], val[2], val[3]);
                }
            }
        }
    }
    };
    auto reset = [&] {
        m_run = -INFINITY; l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    };

    if (RES == 0) {
        const int q0 = blockIdx.x * QT;
        stage_q(q0);
        reset();
        for (int c0 = 0; c0 < Ktot; c0 += KC) {
            __syncthreads();  // previous chunk consumed (and Q image visible on the first pass)
            stage_kv(c0, OFF_K0, OFF_V0);
            const unsigned long long vmask = chunk_mask(c0);
            __syncthreads();
            compute_chunk(OFF_K0, OFF_V0, vmask);
        }
        finish(q0);
    } else {
        unsigned long long vm[NRES];
#pragma unroll
        for (int c = 0; c < NRES; ++c) {
            vm[c] = 0;
            if (c * KC < Ktot) {   // (uniform)
                stage_kv(c * KC, OFF_K0 + c * 2 * K_ARR, OFF_V0 + c * 2 * V_ARR);
                vm[c] = chunk_mask(c * KC);
            }
        }
        for (int qb = 0; qb < N; qb += NG * QT) {   // both groups walk the same number of rounds (barriers are workgroup-wide)
            const int q0 = qb + grp * QT;
            const bool live = q0 < N;               // (uniform per wave)
            __syncthreads();   // the previous round's reads of the Q images are done
            if (live) stage_q(q0);
            reset();
            __syncthreads();   // Q tiles (and, on the first round, every K / V chunk) visible
            if (live) {
#pragma unroll
                for (int c = 0; c < NRES; ++c)
                    if (c * KC < Ktot) compute_chunk(OFF_K0 + c * 2 * K_ARR, OFF_V0 + c * 2 * V_ARR, vm[c]);
                finish(q0);
            }
        }
    }
}

template <int DH, bool PREP, int RES = 0>
static hipError_t attn_mfma_go(const AttnArgs& a, hipStream_t st) {
    constexpr int DHP = DH <= 64 ? 64 : 128;
    constexpr size_t lds = (RES ? 2 : 1) * 2 * (32 * DHP * 2) + (RES ? RES : 1) * (2 * (64 * DHP * 2) + 2 * (DHP * 128));
    static_assert(lds <= 160 * 1024, "resident K / V exceed LDS");
    auto kern = attention_mfma_kernel<DH, PREP, RES>;
    static DevOnce once;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (e != hipSuccess) return e;
    dim3 grid(RES ? 1 : (a.N + 31) / 32, a.H, a.B);
    static const size_t dbg_lds = getenv("SMTTS_DBG_ATTN_LDS") ? (size_t)atol(getenv("SMTTS_DBG_ATTN_LDS")) : 0;
    if (dbg_lds > lds) {   // debug: claim the whole CU's LDS so that no other LDS-using workgroup can share the CU
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dbg_lds);
        AttnArgs b = a; b.dbg_lds_bytes = (int)lds;
        hipLaunchKernelGGL(kern, grid, dim3(RES ? 512 : 256), dbg_lds, st, b);
        return hipGetLastError();
    }
    AttnArgs b = a; b.dbg_lds_bytes = (int)lds;
    hipLaunchKernelGGL(kern, grid, dim3(RES ? 512 : 256), lds, st, b);
    return hipGetLastError();
}

// The resident form pays when the streaming grid would be several rounds and the (batch, head) grid still fills the chip's better
// half: the teacher's CFG batches (B = 24: 576 -> 192 workgroups).  B = 8 (192 -> 64 workgroups) keeps the streaming form.
int g_attn_resident = 1;   // SMTTS_ATTN_RES=0: never (A/B switch, read by the engine)
bool attention_mfma_resident(const AttnArgs& a) {
    const int kt = a.N + (a.k_ref ? a.R : 0) + (a.k_text ? a.P : 0);
    return g_attn_resident && a.dh == 120 && kt <= 128 && a.N > 32 && (long)a.B * a.H >= (g_attn_resident >= 2 ? g_attn_resident : 128);   // (>= 2: A/B override of the grid threshold)
}

// a.prenormed = 1: q, k already RMS-normalised + rotated in place by launch_qk_prep; 0: raw projections, prepared while staging.
// 16-B aligned rows either way.
hipError_t launch_attention_mfma(const AttnArgs& a, hipStream_t st) {
    if (a.N <= 0 || a.B <= 0) return hipSuccess;
    if ((a.rs % 4) || (a.bs % 4) || (a.dh % 4) || (a.ors % 4) || (a.obs % 4)) return hipErrorInvalidValue;
    if (!a.prenormed && ((a.rot_dim % 4) || a.rot_dim > a.dh || !a.qw || !a.kw || (a.rot_dim && (!a.rope_cos || !a.rope_sin))))
        return hipErrorInvalidValue;
    const double kt = a.N + (a.k_ref ? a.R : 0) + (a.k_text ? a.P : 0);
    const double bh = (double)a.B * a.H;
    ProfScope ps(st, a.dh == 120 ? "attention_mfma<120>" : a.dh == 64 ? "attention_mfma<64>" : "attention_mfma<128>",
                 4.0 * bh * a.N * kt * a.dh, 4.0 * bh * a.dh * (5.0 * a.N + 2.0 * (kt - a.N)));
    switch (a.dh) {
        case 64: return a.prenormed ? attn_mfma_go<64, false>(a, st) : attn_mfma_go<64, true>(a, st);
        case 120:
            if (attention_mfma_resident(a)) return a.prenormed ? attn_mfma_go<120, false, 2>(a, st) : attn_mfma_go<120, true, 2>(a, st);
            return a.prenormed ? attn_mfma_go<120, false>(a, st) : attn_mfma_go<120, true>(a, st);
        case 128: return a.prenormed ? attn_mfma_go<128, false>(a, st) : attn_mfma_go<128, true>(a, st);
    }
    return hipErrorInvalidValue;
}
