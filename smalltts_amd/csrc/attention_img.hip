// Joint attention as a DMA + MFMA kernel (round 3), gfx950.
//
// The producers write the attention operands in the kernel's own LDS image format ("the producer writes the consumer's operand
// format"): the QKVG GEMM epilogue (gemm3 EpiQKV) — or the stand-alone qkv_pack kernel below — applies the per-head
// RMSNorm + RoPE + 1/sqrt(dh) of dit.py:95-108 / style.py:21-25,52-55 to its fp32 accumulators and stores
//     Q   [B][H][N][DHP]      K   [B][H][N][DHP]      V^T [B][H][DHP][Np]       sigmoid(gate) [B*N][H*dh]
// as 16-bit operand arrays (one fp16 / bf16 array, or a split-bf16 hi + lo pair), DHP = head dim padded to 64 / 128 with zeros,
// Np = N rounded up to 8 keys (pad columns zero); cross_pack does the same once per sampler call for the cross-KV cache of
// all layers:  Kc [L][B][H][Cp][DHP], Vc^T [L][B][H][DHP][Cp], Cp = pad8(R) + pad8(P).
// This kernel then only moves 16-byte pieces HBM -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, the XOR swizzle applied on
// the per-lane SOURCE address like gemm3) and runs MFMAs: no conversion, no normalisation, no transposition in here.
//
//   key positions of a (batch, head):  [0, Np) self keys | [Np, Np + Rp) reference keys | [Np + Rp, Np + Cp) text keys
//   (every 8-key group of V^T comes from ONE source array, so a 16-byte DMA piece never straddles two)
//   S^T[key][query] = K . Q^T      A = K chunk [64 keys][DHP], B = Q tile [32 queries][DHP]  -> lane = query: softmax is lane-local
//   O^T[dim][query] = V^T . P^T    A = V^T chunk [DHP][64 keys], B = P^T built in registers with v_permlane32_swap
// Workgroup = 4 waves = one (batch, head) and a strided set of 32-query tiles; every wave computes the whole S^T chunk (16 MFMAs
// at fp16) and owns one 32-dim tile of O^T.  All chunks of K / V^T stay resident in LDS when they fit (Ktot <= 128 at fp16 x 4
// slots: the B = 8 and teacher cases), staged ONCE per workgroup; longer key ranges stream through two slots.
#include "kernels.hpp"
#include "prof.hpp"

namespace {

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
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
}

// four consecutive activation values stored by store_split4 (format selected by `lo`)
__device__ __forceinline__ float4 load_act4(const bf16_t* hi, const bf16_t* lo, long off) {
    if (sm_is_f16(lo)) {
        const half4_t h = *reinterpret_cast<const half4_t*>(hi + off);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(hi + off);
    float4 r = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    if (lo) {
        const bf16x4 l = *reinterpret_cast<const bf16x4*>(lo + off);
        r.x += (float)l[0]; r.y += (float)l[1]; r.z += (float)l[2]; r.w += (float)l[3];
    }
    return r;
}

// two probabilities -> one packed 16-bit pair in the operand format (hi) and, for the split format, the residual pair (lo)
template <int SPLIT>
__device__ __forceinline__ void pack_p2(float a, float b, unsigned& hi, unsigned& lo) {
    if constexpr (SPLIT == PREC_F16) {
        f32x2_t v; v.x = a; v.y = b;
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));   // p in [0, 1]: no saturation needed
        lo = 0;
    } else {
        const bf16_t ah = (bf16_t)a, bh = (bf16_t)b;
        hi = (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
        if constexpr (SPLIT == PREC_BF16X3) {
            const bf16_t al = (bf16_t)(a - (float)ah), bl = (bf16_t)(b - (float)bh);
            lo = (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
        } else {
            lo = 0;
        }
    }
}

template <int DHP, int SPLIT>
__global__ __launch_bounds__(256, 2) void attention_img_kernel(AttnImg a, int nslots) {   // (two workgroups per CU: <= 256 registers)
    constexpr int NARR = SPLIT == PREC_BF16X3 ? 2 : 1;
    constexpr int KC = 64, QT = 32;
    constexpr int PITCH = DHP * 2;              // bytes per row of the Q / K images
    constexpr int PPR = PITCH / 16;             // 16-byte pieces per row (8 or 16)
    constexpr int RPI = 64 / PPR;               // rows per DMA wave-instruction (8 or 4)
    constexpr int Q_ARR = QT * PITCH, K_ARR = KC * PITCH, V_ARR = DHP * 128;
    constexpr int QSLOT = NARR * Q_ARR, CSLOT = NARR * (K_ARR + V_ARR);
    constexpr int OFF_C = 2 * QSLOT;            // two Q slots, then the chunk slots
    constexpr int NDT = DHP / 32;               // 32-dim tiles of O^T; wave w owns tile w (w < NDT)
    constexpr int KS1 = DHP / 16;               // k16 steps of S^T
    constexpr int NQI = QT / RPI, NKI = KC / RPI, NVI = DHP / 8;   // DMA instructions per array: Q tile, K chunk, V^T chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = a.N, Np = a.Np, Cp = a.kc ? a.Cp : 0, Rp = a.Rp;
    const int Kpos = Np + Cp, nch = (Kpos + KC - 1) / KC;
    const long bh = (long)b * a.H + h;
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)SM_LPTR(smem);
    const bool resident = nch <= nslots;        // (uniform) every chunk keeps its own slot for the whole workgroup
    // The image / mask pointers as values in SGPRs.  Written as `cond ? a.kc : a.k` with a per-lane condition, hipcc selected the
    // FIELD OFFSET per lane and loaded the pointer from the kernel-argument segment with a vector load — a dependent memory round trip
    // (and an s_waitcnt vmcnt(0) that also drained the DMAs already in flight) in front of every DMA instruction of a chunk, four more in
    // the key-mask chain: ~20 serialised round trips, most of this kernel's 12 us (round 6, found in the ISA).  An empty asm makes each
    // pointer an opaque register value; the per-lane choice is then a v_cndmask.
    auto pin = [](auto* p) { asm volatile("" : "+s"(p)); return p; };
    const bf16_t* const pk[2] = {pin(a.k), NARR == 2 ? pin(a.k_lo) : nullptr};
    const bf16_t* const pkc[2] = {pin(a.kc), NARR == 2 ? pin(a.kc_lo) : nullptr};
    const bf16_t* const pvt[2] = {pin(a.vt), NARR == 2 ? pin(a.vt_lo) : nullptr};
    const bf16_t* const pvtc[2] = {pin(a.vtc), NARR == 2 ? pin(a.vtc_lo) : nullptr};
    const uint8_t* const pms = pin(a.mask_self);
    const uint8_t* const pmr = pin(a.mask_ref);
    const uint8_t* const pmt = pin(a.mask_text);
    const int aR = a.R, aP = a.P;

    auto swz = [](int row, int c) { return PPR == 16 ? (c ^ (row & 15)) : (c ^ ((row >> 1) & 7)); };

    // ---- DMA issue: each wave takes every 4th wave-instruction of a tile; a lane's source = the piece that belongs at its LDS place
    auto issue_q = [&](int q0, int qs) {
#pragma unroll
        for (int ar = 0; ar < NARR; ++ar)
#pragma unroll
            for (int i = 0; i < NQI / 4; ++i) {
                const int ii = i * 4 + w;
                const int row = ii * RPI + lane / PPR, p = lane % PPR;
                int n = q0 + row;
                n = n < N ? n : N - 1;
                const bf16_t* src = (ar ? a.q_lo : a.q) + (bh * N + n) * DHP + swz(row, p) * 8;
                dma16(src, lds0 + (unsigned)(qs * QSLOT + ar * Q_ARR + ii * 1024));
            }
    };
    auto issue_chunk = [&](int c0, int slot) {
        const unsigned base = lds0 + (unsigned)(OFF_C + slot * CSLOT);
#pragma unroll
        for (int ar = 0; ar < NARR; ++ar) {
#pragma unroll
            for (int i = 0; i < NKI / 4; ++i) {
                const int ii = i * 4 + w;
                const int row = ii * RPI + lane / PPR, p = lane % PPR;
                const int kp = c0 + row;
                const bf16_t* src;
                if (kp < Np || Cp == 0) {
                    const int n = kp < N ? kp : N - 1;            // pad rows: any readable row (masked out below)
                    src = pk[ar] + (bh * N + n) * DHP;
                } else {
                    int j = kp - Np;
                    j = j < Cp ? j : Cp - 1;
                    src = pkc[ar] + (bh * Cp + j) * DHP;
                }
                dma16(src + swz(row, p) * 8, base + (unsigned)(ar * K_ARR + ii * 1024));
            }
#pragma unroll
            for (int i = 0; i < NVI / 4; ++i) {
                const int ii = i * 4 + w;
                const int dr = ii * 8 + (lane >> 3), p = lane & 7;
                const int gp = c0 + ((p ^ ((dr >> 1) & 7)) << 3);    // first key position of this 8-key group
                const bf16_t* src;
                if (gp < Np) {
                    src = pvt[ar] + (bh * DHP + dr) * Np + gp;
                } else if (gp < Kpos) {
                    src = pvtc[ar] + (bh * DHP + dr) * Cp + (gp - Np);
                } else {
                    src = pvt[ar] + (bh * DHP + dr) * Np;   // beyond the last key: finite data, P = 0 there
                }
                dma16(src, base + (unsigned)(NARR * K_ARR + ar * V_ARR + ii * 1024));
            }
        }
    };
    // key validity of a chunk as a 64-bit mask (lane = key position), identical in every wave.  ONE byte load per lane: which mask
    // array and which index is a per-lane select of register values, not three branch arms with a load and a wait each
    auto chunk_byte = [&](int c0, bool& inr) -> int {
        const int kp = c0 + lane;
        const uint8_t* mp;
        long mi;
        if (kp < Np) {
            mp = pms; mi = (long)b * N + kp; inr = kp < N;
        } else if (kp < Np + Rp) {
            const int j = kp - Np;
            mp = pmr; mi = (long)b * aR + j; inr = j < aR;
        } else {
            const int j = kp - Np - Rp;
            mp = pmt; mi = (long)b * aP + j; inr = kp < Kpos && j < aP;
        }
        // unconditional load of a readable byte (an exec-masked `if (inr && mp)` put a wait at its join: four serial round trips for
        // the four chunk masks); the value only counts where the key is in range and its mask exists
        const bool use = inr && mp != nullptr;
        const uint8_t* const sp = use ? mp + mi : reinterpret_cast<const uint8_t*>(pk[0]);
        const int v = *sp;
        return use ? v : 1;
    };
    auto chunk_mask = [&](int c0) -> unsigned long long {
        bool inr;
        const int v = chunk_byte(c0, inr);
        return __ballot(inr && v != 0);
    };

    float m_run, l_run;   // per query (lane & 31); both lane halves keep identical copies
    floatx16 oacc;

    // ---- one chunk of keys against the staged query tile: S^T, online softmax, O^T += V^T P^T ---------------------------------
    auto compute_chunk = [&](int qs, int slot, unsigned long long vmask) {
        const int OFF_Q = qs * QSLOT, OFF_K = OFF_C + slot * CSLOT, OFF_V = OFF_K + NARR * K_ARR;
        floatx16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int c = ks * 2 + fh;
            const int qoff = OFF_Q + fr * PITCH + (swz(fr, c) << 4);
            const bf16x8 qh = *reinterpret_cast<const bf16x8*>(smem + qoff);
            bf16x8 ql;
            if (SPLIT == PREC_BF16X3) ql = *reinterpret_cast<const bf16x8*>(smem + qoff + Q_ARR);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kr = t * 32 + fr;
                const int koff = OFF_K + kr * PITCH + (swz(kr, c) << 4);
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(smem + koff);
                if (SPLIT == PREC_BF16X3) {
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(smem + koff + K_ARR);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, s[t], 0, 0, 0);
                }
                s[t] = mfma16<SPLIT>(kh, qh, s[t]);
            }
        }
        // online softmax over keys for this lane's query: key(t, r) = 32 t + (r & 3) + 8 (r >> 2) + 4 fh
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

        if (w < NDT) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int t = ks >> 1, r0 = (ks & 1) * 8;
                unsigned xh[2], yh[2], xl[2], yl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    pack_p2<SPLIT>(s[t][r0 + 2 * e], s[t][r0 + 2 * e + 1], xh[e], xl[e]);
                    pack_p2<SPLIT>(s[t][r0 + 4 + 2 * e], s[t][r0 + 4 + 2 * e + 1], yh[e], yl[e]);
                }
                // half 0 needs keys 0..7 of the step, half 1 keys 8..15: swap the upper half of X with the lower half of Y
                unsigned fh_[4], fl_[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto rh = __builtin_amdgcn_permlane32_swap(xh[e], yh[e], false, false);
                    fh_[e] = rh[0]; fh_[2 + e] = rh[1];
                    if (SPLIT == PREC_BF16X3) {
                        auto rl = __builtin_amdgcn_permlane32_swap(xl[e], yl[e], false, false);
                        fl_[e] = rl[0]; fl_[2 + e] = rl[1];
                    }
                }
                const bf16x8 ph = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fh_));
                const int vr = w * 32 + fr;
                const int voff = OFF_V + vr * 128 + (((ks * 2 + fh) ^ ((vr >> 1) & 7)) << 4);
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(smem + voff);
                if (SPLIT == PREC_BF16X3) {
                    const bf16x8 pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(fl_));
                    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(smem + voff + V_ARR);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc, 0, 0, 0);
                }
                oacc = mfma16<SPLIT>(vh, ph, oacc);
            }
        }
    };

    // ---- normalise, gate, store: lane = query fr, rows = dims 32 w + (r & 3) + 8 (r >> 2) + 4 fh -----------------------------
    // sigmoid(gate) of this lane's output elements: requested at the top of a tile, together with the DMAs, so that the
    // epilogue does not start with a dependent memory round trip
    // (RAW 16-bit words: the conversion waits for the data, so it happens in finish() — converting where the loads are issued put an
    // s_waitcnt vmcnt(0) behind each of the four, i.e. four dependent round trips that also drained the tile's DMAs)
    uint2 graw[4], glraw[4];
    const bf16_t* const pg = pin(a.g);
    const bf16_t* const pgl = SPLIT == PREC_BF16X3 ? pin(a.g_lo) : nullptr;
    auto load_gate = [&](int q0) {
        const int n = q0 + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d0 = 32 * w + 8 * q + 4 * fh;
            graw[q] = make_uint2(0u, 0u);
            if (SPLIT == PREC_BF16X3) glraw[q] = make_uint2(0u, 0u);
            if (w < NDT && n < N && d0 < a.dh) {
                const long off = ((long)b * N + n) * ((long)a.H * a.dh) + h * a.dh + d0;
                graw[q] = *reinterpret_cast<const uint2*>(pg + off);
                if (SPLIT == PREC_BF16X3) glraw[q] = *reinterpret_cast<const uint2*>(pgl + off);
            }
        }
    };
    auto gate4 = [&](int q) -> float4 {   // what store_split4 wrote in the format of this instantiation
        if constexpr (SPLIT == PREC_F16) {
            const half2_t a0 = __builtin_bit_cast(half2_t, graw[q].x), a1 = __builtin_bit_cast(half2_t, graw[q].y);
            return make_float4((float)a0[0], (float)a0[1], (float)a1[0], (float)a1[1]);
        } else {
            float4 r = make_float4(__uint_as_float(graw[q].x << 16), __uint_as_float(graw[q].x & 0xffff0000u),
                                   __uint_as_float(graw[q].y << 16), __uint_as_float(graw[q].y & 0xffff0000u));
            if constexpr (SPLIT == PREC_BF16X3) {
                r.x += __uint_as_float(glraw[q].x << 16); r.y += __uint_as_float(glraw[q].x & 0xffff0000u);
                r.z += __uint_as_float(glraw[q].y << 16); r.w += __uint_as_float(glraw[q].y & 0xffff0000u);
            }
            return r;
        }
    };
    auto finish = [&](int q0) {
        const int n = q0 + fr;
        if (w < NDT && n < N) {
            const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
            const long obase = ((long)b * N + n) * a.ors + h * a.dh;
            float4 g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // the (conditional) gate loads are complete before the first store (gemm.hpp epi_settle)
                asm volatile("" : "+v"(graw[q].x), "+v"(graw[q].y));
                if (SPLIT == PREC_BF16X3) asm volatile("" : "+v"(glraw[q].x), "+v"(glraw[q].y));
                g4[q] = gate4(q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d0 = 32 * w + 8 * q + 4 * fh;
                if (d0 < a.dh)   // dh % 4 == 0: a group of four dims is entirely inside or outside the head
                    store_split4(a.out_hi, a.out_lo, obase + d0,
                                 make_float4(oacc[4 * q + 0] * inv * g4[q].x, oacc[4 * q + 1] * inv * g4[q].y,
                                             oacc[4 * q + 2] * inv * g4[q].z, oacc[4 * q + 3] * inv * g4[q].w));
            }
        }
    };
    auto reset = [&] {
        m_run = -INFINITY; l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    };

    const int ntiles = (N + QT - 1) / QT;
    int qt = blockIdx.x;
    if (qt >= ntiles) return;   // (uniform)
    issue_q(qt * QT, 0);
    // resident: every chunk has its own slot — staged ONCE per workgroup, all DMAs in flight together (one memory round trip);
    // otherwise the chunks stream through two slots, chunk c + 1 in flight while chunk c is consumed
    if (resident)
        for (int c = 0; c < nch; ++c) issue_chunk(c * KC, c);
    unsigned long long vm0 = 0, vm1 = 0, vm2 = 0, vm3 = 0;   // chunk masks, loaded once when there are at most four chunks
    if (nch <= 4) {   // (all four byte loads go out before the first ballot looks at one)
        bool i0 = false, i1 = false, i2 = false, i3 = false;
        const int v0 = chunk_byte(0, i0);
        const int v1 = chunk_byte(KC, i1);          // (chunks past the last: every key out of range, the load reads a dummy byte)
        const int v2 = chunk_byte(2 * KC, i2);
        const int v3 = chunk_byte(3 * KC, i3);
        vm0 = __ballot(i0 && v0 != 0);
        vm1 = __ballot(i1 && v1 != 0);
        vm2 = __ballot(i2 && v2 != 0);
        vm3 = __ballot(i3 && v3 != 0);
    }
    for (int it = 0; qt < ntiles; qt += gridDim.x, ++it) {
        if (!resident) {
            if (it) issue_q(qt * QT, it & 1);   // (slot last read two tiles ago)
            issue_chunk(0, 0);
        }
        reset();
        load_gate(qt * QT);
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const unsigned long long vmask = nch <= 4 ? (c == 0 ? vm0 : c == 1 ? vm1 : c == 2 ? vm2 : vm3) : chunk_mask(c * KC);
            if (c == 0 || !resident) {
                wait_vmcnt<0>();
                __syncthreads();   // Q tile + chunk c landed for every wave; every wave is done with the previous chunk / tile
                if (resident) {
                    if (qt + (int)gridDim.x < ntiles) issue_q((qt + gridDim.x) * QT, (it + 1) & 1);
                } else if (c + 1 < nch) {
                    issue_chunk((c + 1) * KC, (c + 1) & 1);
                }
            }
            compute_chunk(it & 1, resident ? c : (c & 1), vmask);
        }
        finish(qt * QT);
        if (!resident) __syncthreads();   // slot 0 is re-filled by the next tile's first chunk
    }
}

template <int DHP, int SPLIT>
hipError_t attn_img_go(const AttnImg& a, hipStream_t st) {
    constexpr int NARR = SPLIT == PREC_BF16X3 ? 2 : 1;
    constexpr size_t qslot = (size_t)NARR * 32 * DHP * 2, cslot = (size_t)NARR * (64 * DHP * 2 + DHP * 128);
    const int Cp = a.kc ? a.Cp : 0;
    const int nch = (a.Np + Cp + 63) / 64;
    const int max_slots = (int)((160 * 1024 - 2 * qslot) / cslot);
    int nslots = nch <= max_slots && nch <= 4 ? nch : 2;
    const size_t lds = 2 * qslot + (size_t)nslots * cslot;
    auto kern = attention_img_kernel<DHP, SPLIT>;
    static DevOnce once;
    int cus = 256;
    hipError_t e = once.ensure([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }, &cus);
    if (e != hipSuccess) return e;
    // query tiles per workgroup: one while (batch x heads x tiles) fits the chip about once, else the keys are staged once per
    // (batch, head) and the workgroup walks its tiles (the teacher's 3B-row CFG batches)
    const int ntiles = (a.N + 31) / 32;
    int gx = ntiles;
    if (nslots == nch)
        while (gx > 1 && (long)gx * a.H * a.B > 2L * cus) gx = (gx + 1) / 2;
    dim3 grid(gx, a.H, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, nslots);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_img(const AttnImg& a, hipStream_t st) {
    if (a.N <= 0 || a.B <= 0) return hipSuccess;
    if ((a.dh % 4) || (a.ors % 4) || (a.Np % 8) || (a.Cp % 8) || (a.Rp % 8) || a.Np < a.N) return hipErrorInvalidValue;
    const double kt = a.N + (a.kc ? a.R + a.P : 0);
    const double bhd = (double)a.B * a.H * a.dh;
    const double eb = a.prec == PREC_BF16X3 ? 4.0 : 2.0;   // bytes per operand element
    ProfScope ps(st, a.dh <= 64 ? "attention_img<64>" : "attention_img<128>",   // (padded head dim: the kernel's template argument)
                 4.0 * bhd * a.N * kt, eb * bhd * (4.0 * a.N + 2.0 * (kt - a.N)));
    const int dhp = a.dh <= 64 ? 64 : 128;
    if (a.dh > 128) return hipErrorInvalidValue;
    switch (a.prec) {
        case PREC_BF16X3: return dhp == 64 ? attn_img_go<64, PREC_BF16X3>(a, st) : attn_img_go<128, PREC_BF16X3>(a, st);
        case PREC_F16: return dhp == 64 ? attn_img_go<64, PREC_F16>(a, st) : attn_img_go<128, PREC_F16>(a, st);
        case PREC_BF16: return dhp == 64 ? attn_img_go<64, PREC_BF16>(a, st) : attn_img_go<128, PREC_BF16>(a, st);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Producers outside the GEMM: the stand-alone q / k / v / gate pack (test hook and reference for the gemm3 EpiQKV epilogue) and the
// cross-KV pack (once per sampler call).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// one wave per (row m = (b, n), head h); lane -> dims 2 lane, 2 lane + 1 (a rotation pair)
__global__ __launch_bounds__(256) void qkv_pack_kernel(QkvPackArgs p) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)p.B * p.N * p.H;
    if (wid >= total) return;
    const int h = (int)(wid % p.H);
    const long m = wid / p.H;
    const int n = (int)(m % p.N), b = (int)(m / p.N);
    const int dh = p.dh, DHP = p.dhp, d0 = 2 * lane;
    const long D = (long)p.H * dh;
    const float* row = p.qkvg + m * 4 * D + (long)h * dh;
    const bool in = d0 < dh;        // dh is even: both dims of the pair are inside or outside
    float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f, g0 = 0.f, g1 = 0.f;
    if (in) {
        q0 = row[d0]; q1 = row[d0 + 1];
        k0 = row[D + d0]; k1 = row[D + d0 + 1];
        v0 = row[2 * D + d0]; v1 = row[2 * D + d0 + 1];
        g0 = row[3 * D + d0]; g1 = row[3 * D + d0 + 1];
    }
    // the arithmetic of the gemm3 EpiQKV epilogue (common.hpp QkPrep), value for value
    const float qs = wave_sum(fmaf(q1, q1, q0 * q0)), ks = wave_sum(fmaf(k1, k1, k0 * k0));
    const QkPrep pq{p.qw, p.rope_cos, p.rope_sin, p.rot_dim, dh, p.eps, p.q_scale}, pk{p.kw, p.rope_cos, p.rope_sin, p.rot_dim, dh, p.eps, 1.0f};
    pq.apply(q0, q1, qs, h, d0, n);
    pk.apply(k0, k1, ks, h, d0, n);
    if (d0 < DHP) {
        const long bhh = (long)b * p.H + h;
        const long qo = (bhh * p.N + n) * DHP + d0;
        store_img2(p.q, p.q_lo, p.prec, qo, q0, q1);
        store_img2(p.k, p.k_lo, p.prec, qo, k0, k1);
        const long vo = (bhh * DHP + d0) * p.Np + n;
        store_img1(p.vt, p.vt_lo, p.prec, vo, v0); store_img1(p.vt, p.vt_lo, p.prec, vo + p.Np, v1);
        if (in) {
            const long go = m * D + (long)h * dh + d0;
            store_img2(p.g, p.g_lo, p.prec, go, sigmoid_f(g0), sigmoid_f(g1));
        }
    }
}

// thread -> (l, b, h, j, d): Kc[l][b][h][j][d] and Vc^T[l][b][h][d][j] from the fp32 rank-5 caches; pad rows / dims are zero
__global__ __launch_bounds__(256) void cross_pack_kernel(CrossPackArgs p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.L * p.B * p.H * p.Cp * p.dhp;
    if (i >= total) return;
    const int d = (int)(i % p.dhp);
    long t = i / p.dhp;
    const int j = (int)(t % p.Cp);
    t /= p.Cp;                       // t = (l * B + b) * H + h
    float kv = 0.f, vv = 0.f;
    if (d < p.dh) {
        if (j < p.R) {
            const long o = (t * p.R + j) * p.dh + d;
            kv = p.k_ref[o]; vv = p.v_ref[o];
        } else if (j >= p.Rp && j - p.Rp < p.P) {
            const long o = (t * p.P + (j - p.Rp)) * p.dh + d;
            kv = p.k_text[o]; vv = p.v_text[o];
        }
    }
    store_img1(p.kc, p.kc_lo, p.prec, i, kv);
    store_img1(p.vtc, p.vtc_lo, p.prec, (t * p.dhp + d) * p.Cp + j, vv);
}

}  // namespace

hipError_t launch_qkv_pack(const QkvPackArgs& p, hipStream_t st) {
    const long total = (long)p.B * p.N * p.H;
    if (total == 0) return hipSuccess;
    if ((p.dh & 1) || p.dhp > 128 || (p.rot_dim & 1) || p.rot_dim > p.dh) return hipErrorInvalidValue;
    ProfScope ps(st, "qkv_pack", 16.0 * total * p.dh, 24.0 * total * p.dh);
    hipLaunchKernelGGL(qkv_pack_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_cross_pack(const CrossPackArgs& p, hipStream_t st) {
    const long total = (long)p.L * p.B * p.H * p.Cp * p.dhp;
    if (total == 0) return hipSuccess;
    ProfScope ps(st, "cross_pack", 0.0, 12.0 * total);
    hipLaunchKernelGGL(cross_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

namespace {
__global__ void split_to_f32_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)hi[i] + (float)lo[i];
}
}  // namespace
hipError_t launch_split_to_f32(const bf16_t* hi, const bf16_t* lo, float* out, long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hi, lo, out, n);
    return hipGetLastError();
}
