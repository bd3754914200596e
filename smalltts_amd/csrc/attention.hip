// Fused joint attention for the smalltts DiT and the two condition encoders (gfx950).
//
// Reference semantics (dit.py:95-135, style.py:48-68):
//   q = RoPE(RMSNorm_head(q_raw) ), k_self = RoPE(RMSNorm_head(k_raw)), keys = [k_self | k_ref | k_text]
//   o = softmax(q keys^T / sqrt(dh) + keymask) [v_self | v_ref | v_text] ;  out = o * sigmoid(gate)
// Cross keys arrive already normalised from the KV cache and carry no RoPE (dit.py:83-86).
//
// Work decomposition: one workgroup per (16-query tile, head, batch row) -> >=300 workgroups at the
// bench shape; K/V are staged through LDS in 64-key chunks with an online softmax, so Ktot is
// unbounded.  Inside a chunk: lane = key for q.k^T (ds_read_b128 rows, padded stride -> no bank
// conflicts), wave-wide max/sum by cross-lane shuffles, then lane = head-dim for P.V with P
// broadcast from LDS.  All math fp32: attention is 0.16 % of the model FLOPs and
// its scores need fp32 to hold the 1e-3 latent parity.
#include "kernels.hpp"
#include "prof.hpp"

template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    constexpr int KC = 64, QW = 4, QT = 16;
    constexpr int KS = DH + 4;             // K row stride in LDS (floats)
    constexpr int DPL = (DH + 63) / 64;    // head dims per lane in the P.V phase
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* K_s = smem;                     // [KC][KS]
    float* V_s = K_s + KC * KS;            // [KC][DH]
    float* q_s = V_s + KC * DH;            // [QT][DH]
    float* p_s = q_s + QT * DH;            // [4 waves][KC][QW]
    float* m_s = p_s + 4 * KC * QW;        // [KC] key validity

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int N = a.N, R = a.k_ref ? a.R : 0, P = a.k_text ? a.P : 0;
    const int Ktot = N + R + P;
    const float inv_dh = 1.0f / (float)DH;
    const float sm_scale = 1.0f / sqrtf((float)DH);

    // RMSNorm (per-head weight) + RoPE on interleaved pairs for one row held as DPL regs per lane
    auto norm_rope = [&](const float* src, const float* wgt, int pos, float out[DPL]) {
        float x[DPL];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            int d = lane + 64 * i;
            x[i] = d < DH ? src[d] : 0.f;
            ss += x[i] * x[i];
        }
        ss = wave_sum(ss);
        const float rstd = 1.0f / sqrtf(ss * inv_dh + a.eps);
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            int d = lane + 64 * i;
            float y = d < DH ? x[i] * rstd * wgt[d] : 0.f;
            float partner = __shfl_xor(y, 1, 64);
            if (d < a.rot_dim) {
                const float c = a.rope_cos[(long)pos * a.rot_dim + d], s = a.rope_sin[(long)pos * a.rot_dim + d];
                y = (d & 1) ? (y * c + partner * s) : (y * c - partner * s);
            }
            out[i] = y;
        }
    };

    // ---- queries of this wave -> LDS (pre-scaled by 1/sqrt(dh)) ------------------------------
    const int q0 = qt * QT + w * QW;
    float qpre[QW][DPL];
    if (a.prenormed) {
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            const int n = q0 + qi < N ? q0 + qi : N - 1;
            const float* qp = a.q + (long)b * a.bs + (long)n * a.rs + h * DH;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                qpre[qi][i] = qp[d < DH ? d : DH - 1];
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        const int n = q0 + qi;
        float qv[DPL];
        if (n < N && a.prenormed) {
#pragma unroll
            for (int i = 0; i < DPL; ++i) qv[i] = qpre[qi][i];
        } else if (n < N) {
            norm_rope(a.q + (long)b * a.bs + (long)n * a.rs + h * DH, a.qw + h * DH, n, qv);
        } else {
#pragma unroll
            for (int i = 0; i < DPL; ++i) qv[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            int d = lane + 64 * i;
            if (d < DH) q_s[(w * QW + qi) * DH + d] = qv[i] * sm_scale;
        }
    }

    float m_run[QW], l_run[QW], o[QW][DPL];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        m_run[qi] = -INFINITY;
        l_run[qi] = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) o[qi][i] = 0.f;
    }

    for (int c0 = 0; c0 < Ktot; c0 += KC) {
        __syncthreads();  // previous chunk fully consumed (also orders the q_s writes on entry)
        // ---- stage K/V chunk: each wave fills rows w, w+4, ... ------------------------------
        if (a.prenormed) {
            // keys are plain copies: issue the loads of all 16 rows of this wave first, then the LDS stores
            constexpr int RPW = KC / 4;
            float kr_[RPW][DPL], vr_[RPW][DPL];
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int gk = c0 + w + 4 * t;
                const int gc = gk < Ktot ? gk : Ktot - 1;  // clamp: always a valid row, masked out below
                const float* kp;
                const float* vp;
                if (gc < N) {
                    const long base = (long)b * a.bs + (long)gc * a.rs + h * DH;
                    kp = a.k + base; vp = a.v + base;
                } else if (gc < N + R) {
                    const long base = (((long)b * a.H + h) * R + (gc - N)) * DH;
                    kp = a.k_ref + base; vp = a.v_ref + base;
                } else {
                    const long base = (((long)b * a.H + h) * P + (gc - N - R)) * DH;
                    kp = a.k_text + base; vp = a.v_text + base;
                }
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    const int d = lane + 64 * i;
                    const int dc = d < DH ? d : DH - 1;
                    kr_[t][i] = kp[dc];
                    vr_[t][i] = vp[dc];
                }
            }
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int jj = w + 4 * t;
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    const int d = lane + 64 * i;
                    if (d < DH) {
                        K_s[jj * KS + d] = kr_[t][i];
                        V_s[jj * DH + d] = vr_[t][i];
                    }
                }
            }
            // key validity: one lane per key (a consumed 1-byte load per row would drain the whole load queue)
            if (tid < KC) {
                const int gk = c0 + tid;
                float ok = 0.f;
                if (gk < Ktot) {
                    const uint8_t* mk = gk < N ? a.mask_self : (gk < N + R ? a.mask_ref : a.mask_text);
                    const int mi = gk < N ? b * N + gk : (gk < N + R ? b * R + (gk - N) : b * P + (gk - N - R));
                    ok = (!mk || mk[mi]) ? 1.f : 0.f;
                }
                m_s[tid] = ok;
            }
        } else
        for (int jj = w; jj < KC; jj += 4) {
            const int gk = c0 + jj;
            float kv[DPL], vv[DPL];
            float valid = 0.f;
            if (gk < N) {
                const long base = (long)b * a.bs + (long)gk * a.rs + h * DH;
                norm_rope(a.k + base, a.kw + h * DH, gk, kv);
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    int d = lane + 64 * i;
                    vv[i] = d < DH ? a.v[base + d] : 0.f;
                }
                valid = (!a.mask_self || a.mask_self[b * N + gk]) ? 1.f : 0.f;
            } else if (gk < Ktot) {
                const bool isref = gk < N + R;
                const int j = isref ? gk - N : gk - N - R;
                const int S = isref ? R : P;
                const long base = (((long)b * a.H + h) * S + j) * DH;
                const float* kp = (isref ? a.k_ref : a.k_text) + base;
                const float* vp = (isref ? a.v_ref : a.v_text) + base;
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    int d = lane + 64 * i;
                    kv[i] = d < DH ? kp[d] : 0.f;
                    vv[i] = d < DH ? vp[d] : 0.f;
                }
                const uint8_t* mk = isref ? a.mask_ref : a.mask_text;
                valid = (!mk || mk[b * S + j]) ? 1.f : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < DPL; ++i) { kv[i] = 0.f; vv[i] = 0.f; }
            }
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                int d = lane + 64 * i;
                if (d < DH) {
                    K_s[jj * KS + d] = kv[i];
                    V_s[jj * DH + d] = vv[i];
                }
            }
            if (lane == 0) m_s[jj] = valid;
        }
        __syncthreads();

        // ---- scores: lane = key --------------------------------------------------------------
        float s[QW];
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) s[qi] = 0.f;
        const float4* kr = reinterpret_cast<const float4*>(K_s + lane * KS);
#pragma unroll 6
        for (int d4 = 0; d4 < DH / 4; ++d4) {
            const float4 kk = kr[d4];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                const float4 qq = reinterpret_cast<const float4*>(q_s + (w * QW + qi) * DH)[d4];
                s[qi] += qq.x * kk.x + qq.y * kk.y + qq.z * kk.z + qq.w * kk.w;
            }
        }
        const bool kvalid = m_s[lane] != 0.f;
        float p4[QW], sv[QW], cm[QW], al[QW], cs[QW];
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) cm[qi] = sv[qi] = kvalid ? s[qi] : -INFINITY;
        // the four queries' cross-lane reductions advance in lock-step: one wait per butterfly step, not per query
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float t[QW];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) t[qi] = __shfl_xor(cm[qi], off, 64);
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) cm[qi] = fmaxf(cm[qi], t[qi]);
        }
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            const float m_new = fmaxf(m_run[qi], cm[qi]);
            float alpha = 1.f, p = 0.f;
            if (m_new != -INFINITY) {
                alpha = (m_run[qi] == -INFINITY) ? 0.f : __expf(m_run[qi] - m_new);
                p = kvalid ? __expf(sv[qi] - m_new) : 0.f;
            }
            al[qi] = alpha;
            m_run[qi] = m_new;
            p4[qi] = cs[qi] = p;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float t[QW];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) t[qi] = __shfl_xor(cs[qi], off, 64);
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) cs[qi] += t[qi];
        }
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            l_run[qi] = l_run[qi] * al[qi] + cs[qi];
#pragma unroll
            for (int i = 0; i < DPL; ++i) o[qi][i] *= al[qi];
        }
        float* pw = p_s + w * KC * QW;
        *reinterpret_cast<float4*>(pw + lane * QW) = make_float4(p4[0], p4[1], p4[2], p4[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- P.V: lane = head dim ------------------------------------------------------------
        const int kc = (Ktot - c0) < KC ? (Ktot - c0) : KC;
#pragma unroll 8
        for (int jj = 0; jj < kc; ++jj) {
            const float4 pp = *reinterpret_cast<const float4*>(pw + jj * QW);
            float vv[DPL];
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                int d = lane + 64 * i;
                vv[i] = d < DH ? V_s[jj * DH + d] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                o[0][i] += pp.x * vv[i];
                o[1][i] += pp.y * vv[i];
                o[2][i] += pp.z * vv[i];
                o[3][i] += pp.w * vv[i];
            }
        }
    }

    // ---- normalise, gate, store (all gate loads issued before the first store) ----------------------
    {
        float gv[QW][DPL];
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            const int n = q0 + qi;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                gv[qi][i] = (n < N && d < DH) ? a.gate[(long)b * a.bs + (long)n * a.rs + h * DH + d] : 0.f;
            }
        }
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            const int n = q0 + qi;
            if (n >= N) continue;
            const float inv = l_run[qi] > 0.f ? 1.0f / l_run[qi] : 0.f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                if (d < DH) {
                    const float val = o[qi][i] * inv * sigmoid_f(gv[qi][i]);
                    const long oo = (long)b * a.obs + (long)n * a.ors + h * DH + d;
                    if (a.out_hi) {
                        store_act1(a.out_hi, a.out_lo, oo, val);
                    } else {
                        a.out[oo] = val;
                    }
                }
            }
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void qk_prep_kernel(AttnArgs a) {
    constexpr int DPL = (DH + 63) / 64;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)a.B * a.N * a.H * 2;
    if (wid >= total) return;
    const int which = (int)(wid & 1);
    long t = wid >> 1;
    const int h = (int)(t % a.H); t /= a.H;
    const int n = (int)(t % a.N);
    const int b = (int)(t / a.N);
    float* p = const_cast<float*>(which ? a.k : a.q) + (long)b * a.bs + (long)n * a.rs + h * DH;
    const float* wgt = (which ? a.kw : a.qw) + h * DH;
    float x[DPL];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        int d = lane + 64 * i;
        x[i] = d < DH ? p[d] : 0.f;
        ss += x[i] * x[i];
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)DH + a.eps);
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        int d = lane + 64 * i;
        float y = d < DH ? x[i] * rstd * wgt[d] : 0.f;
        const float partner = __shfl_xor(y, 1, 64);
        if (d < a.rot_dim) {
            const float c = a.rope_cos[(long)n * a.rot_dim + d], s = a.rope_sin[(long)n * a.rot_dim + d];
            y = (d & 1) ? (y * c + partner * s) : (y * c - partner * s);
        }
        if (d < DH) p[d] = y;
    }
}

hipError_t launch_qk_prep(const AttnArgs& a, hipStream_t st) {
    const long total = (long)a.B * a.N * a.H * 2;
    if (total == 0) return hipSuccess;
    ProfScope ps(st, "qk_prep", 8.0 * total * a.dh, 8.0 * total * a.dh);
    dim3 grid((unsigned)((total + 3) / 4));
    switch (a.dh) {
        case 64: hipLaunchKernelGGL(qk_prep_kernel<64>, grid, dim3(256), 0, st, a); break;
        case 120: hipLaunchKernelGGL(qk_prep_kernel<120>, grid, dim3(256), 0, st, a); break;
        case 128: hipLaunchKernelGGL(qk_prep_kernel<128>, grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int DH>
static hipError_t attention_go(const AttnArgs& a, hipStream_t st) {
    constexpr int KC = 64, QW = 4, QT = 16, KS = DH + 4;
    size_t lds = sizeof(float) * (KC * KS + KC * DH + QT * DH + 4 * KC * QW + KC);
    auto kern = attention_kernel<DH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((a.N + QT - 1) / QT, a.H, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t st) {
    if (a.N <= 0 || a.B <= 0) return hipSuccess;
    if (a.rot_dim & 1 || a.rot_dim > a.dh) return hipErrorInvalidValue;
    const double kt = a.N + (a.k_ref ? a.R : 0) + (a.k_text ? a.P : 0);
    const double bh = (double)a.B * a.H;
    // algorithmic: QK^T + PV flops; bytes = q,k,v,gate,out rows once + cross K/V once
    ProfScope ps(st, a.dh == 120 ? "attention<120>" : a.dh == 64 ? "attention<64>" : "attention<128>",
                 4.0 * bh * a.N * kt * a.dh, 4.0 * bh * a.dh * (5.0 * a.N + 2.0 * (kt - a.N)));
    switch (a.dh) {
        case 64: return attention_go<64>(a, st);
        case 120: return attention_go<120>(a, st);
        case 128: return attention_go<128>(a, st);
    }
    return hipErrorInvalidValue;
}
