// Non-GEMM kernels: norms, element-wise sampler steps, RNG, weight synthesis/packing helpers.
// HBM-bound streaming kernels: one wave (or sub-wave group) per row, float4 accesses where the
// layout permits, grid sized to cover the chip (>= 256 CUs x several waves).
#include <cstdlib>
#include "kernels.hpp"
#include "prof.hpp"

#define LAUNCH_CHECK() return hipGetLastError()

// ------------------------------------------------------------------------------------------
// AdaLN:  y = LN(x) * (1 + scale) + shift       one wave per row, row kept in registers
// ------------------------------------------------------------------------------------------
// One wave per row; each lane owns NV4 float4 columns (c4 = lane + 64*i).  x, shift and scale are all requested
// before the first reduction so the row pays one memory round trip, and outputs go out as 8/16-byte stores.
template <int NV4>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo, int M,
                                                          int C, float eps, const float* __restrict__ shift,
                                                          const float* __restrict__ scale, long mod_ld,
                                                          int mod_row0, int mod_rstride, int rpb) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int C4 = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * C);
    const long r = (long)(mod_row0 + (row / rpb) * mod_rstride) * mod_ld;
    const float4* sh4 = reinterpret_cast<const float4*>(shift + r);
    const float4* sc4 = reinterpret_cast<const float4*>(scale + r);
    float4 v[NV4], sh[NV4], sc[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = lane + 64 * i;
        const int cc = c < C4 ? c : 0;
        v[i] = xr[cc];
        sh[i] = sh4[cc];
        sc[i] = sc4[cc];
        if (c >= C4) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        if (lane + 64 * i < C4) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {   // (complete before the first store: see rmsnorm_kernel)
        asm volatile("" : "+v"(sh[i].x), "+v"(sh[i].y), "+v"(sh[i].z), "+v"(sh[i].w));
        asm volatile("" : "+v"(sc[i].x), "+v"(sc[i].y), "+v"(sc[i].z), "+v"(sc[i].w));
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 o = make_float4((v[i].x - mean) * rstd * (1.0f + sc[i].x) + sh[i].x,
                                         (v[i].y - mean) * rstd * (1.0f + sc[i].y) + sh[i].y,
                                         (v[i].z - mean) * rstd * (1.0f + sc[i].z) + sh[i].z,
                                         (v[i].w - mean) * rstd * (1.0f + sc[i].w) + sh[i].w);
            if (yhi) store_split4(yhi, ylo, (long)row * C + c * 4, o);
            else reinterpret_cast<float4*>(y + (long)row * C)[c] = o;
        }
    }
}

hipError_t launch_ln_modulate(const float* x, float* y, bf16_t* yhi, bf16_t* ylo, int M, int C, float eps,
                              const float* shift, const float* scale, long mod_ld, int mod_row0, int mod_rstride,
                              int rows_per_batch, hipStream_t st) {
    if (C > 1024 || C % 4 || mod_ld % 4) return hipErrorInvalidValue;
    ProfScope ps(st, "ln_modulate", 8.0 * M * C, 8.0 * M * C);
    dim3 grid((M + 3) / 4), block(256);
    hipLaunchKernelGGL(ln_modulate_kernel<4>, grid, block, 0, st, x, y, yhi, ylo, M, C, eps, shift, scale, mod_ld,
                       mod_row0, mod_rstride, rows_per_batch);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// RMSNorm with 1-D weight; LPR lanes cooperate on one row (float4 per lane per step)
// ------------------------------------------------------------------------------------------
template <int LPR, int NV4>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, RowMap xmap, float* __restrict__ y,
                                                      bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo, RowMap ymap,
                                                      int M, int C, float eps,
                                                      const float* __restrict__ w) {
    constexpr int RPB = 256 / LPR;  // rows per block
    const int sub = threadIdx.x % LPR;
    const int row = blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < M;
    const int c4n = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + xmap.at(live ? row : 0));
    float4 v[NV4], gw[NV4];   // the weights are requested with the row: a load inside the store loop below would cost one
    float ss = 0.f;           // s_waitcnt vmcnt(0) — a drained store queue — per channel group (loads and stores share the counter)
    const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        int c = sub + LPR * i;
        v[i] = (live && c < c4n) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        gw[i] = (live && c < c4n) ? w4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        asm volatile("" : "+v"(gw[i].x), "+v"(gw[i].y), "+v"(gw[i].z), "+v"(gw[i].w));
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    ss = group_sum<LPR>(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)C + eps);
    if (!live) return;
    const long yo = ymap.at(row);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        int c = sub + LPR * i;
        if (c < c4n) {
            float4 g = gw[i];
            const float4 o = make_float4(v[i].x * rstd * g.x, v[i].y * rstd * g.y, v[i].z * rstd * g.z, v[i].w * rstd * g.w);
            if (yhi) store_split4(yhi, ylo, yo + c * 4, o);
            else reinterpret_cast<float4*>(y + yo)[c] = o;
        }
    }
}

template <int LPR, int NV4>
static hipError_t rmsnorm_go(const float* x, RowMap xmap, float* y, bf16_t* yhi, bf16_t* ylo, RowMap ymap, int M, int C,
                             float eps, const float* w, hipStream_t st) {
    constexpr int RPB = 256 / LPR;
    hipLaunchKernelGGL((rmsnorm_kernel<LPR, NV4>), dim3((M + RPB - 1) / RPB), dim3(256), 0, st, x, xmap, y, yhi, ylo,
                       ymap, M, C, eps, w);
    LAUNCH_CHECK();
}

hipError_t launch_rmsnorm(const float* x, RowMap xmap, float* y, bf16_t* yhi, bf16_t* ylo, RowMap ymap, int M, int C,
                          float eps, const float* w, hipStream_t st) {
    if (C % 4) return hipErrorInvalidValue;
    ProfScope ps(st, "rmsnorm", 4.0 * M * C, 8.0 * M * C);
    int c4 = C / 4;
    if (c4 <= 8) return rmsnorm_go<8, 1>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 16) return rmsnorm_go<16, 1>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 32) return rmsnorm_go<32, 1>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 64) return rmsnorm_go<64, 1>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 128) return rmsnorm_go<64, 2>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 256) return rmsnorm_go<64, 4>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    if (c4 <= 512) return rmsnorm_go<64, 8>(x, xmap, y, yhi, ylo, ymap, M, C, eps, w, st);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// cross-K per-head RMSNorm, in place:  k[L][B][H][S][dh] *= rsqrt(mean_d k^2 + eps) * w[L][H][dh]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void headnorm_kernel(float* __restrict__ k, long rows, int B, int H, int S, int dh,
                                                       float eps, const float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* kr = k + row * dh;
    const int h = (int)((row / S) % H);
    const int l = (int)(row / ((long)S * H * B));
    const float* wr = w + ((long)l * H + h) * dh;
    float v0 = lane < dh ? kr[lane] : 0.f;
    float v1 = lane + 64 < dh ? kr[lane + 64] : 0.f;
    float ss = wave_sum(v0 * v0 + v1 * v1);
    float rstd = 1.0f / sqrtf(ss / (float)dh + eps);
    if (lane < dh) kr[lane] = v0 * rstd * wr[lane];
    if (lane + 64 < dh) kr[lane + 64] = v1 * rstd * wr[lane + 64];
}

hipError_t launch_headnorm(float* k, int L, int B, int H, int S, int dh, float eps, const float* w, hipStream_t st) {
    if (dh > 128) return hipErrorInvalidValue;
    long rows = (long)L * B * H * S;
    if (rows == 0) return hipSuccess;
    ProfScope ps(st, "headnorm", 4.0 * rows * dh, 8.0 * rows * dh);
    hipLaunchKernelGGL(headnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, k, rows, B, H, S, dh, eps, w);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// small element-wise kernels
// ------------------------------------------------------------------------------------------
__global__ void embedding_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                 float* __restrict__ out, int M, int C4, int vocab) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * C4) return;
    int m = (int)(i / C4), c = (int)(i % C4);
    long id = ids[m];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table)[id * C4 + c];
}
hipError_t launch_embedding(const int64_t* ids, const float* table, float* out, int M, int C, int vocab, hipStream_t st) {
    long n = (long)M * (C / 4);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(embedding_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, table, out, M, C / 4, vocab);
    LAUNCH_CHECK();
}

__global__ void time_sinusoid_kernel(const float* __restrict__ t, float* __restrict__ e, int rows) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 256) return;
    int r = i >> 8, c = i & 255;
    int j = c & 127;
    // model.py:25-27 : exp(arange(half).float() * -(ln 1e4 / (half-1))), then 1e3 * t * f
    const float neg = -0.072522365133670733f;  // -ln(1e4)/127
    float f = expf((float)j * neg);
    float a = (1e3f * t[r]) * f;
    e[i] = c < 128 ? sinf(a) : cosf(a);
}
hipError_t launch_time_sinusoid(const float* t, float* e, int rows, hipStream_t st) {
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3((rows * 256 + 255) / 256), dim3(256), 0, st, t, e, rows);
    LAUNCH_CHECK();
}

__global__ void len_mask_kernel(const int64_t* __restrict__ len, uint8_t* __restrict__ mask, int B, int R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * R) return;
    int b = i / R, j = i % R;
    long l = len[b];
    l = l > R ? R : l;
    mask[i] = j < l ? 1 : 0;
}
hipError_t launch_len_mask(const int64_t* len, uint8_t* mask, int B, int R, hipStream_t st) {
    if (B * R == 0) return hipSuccess;
    hipLaunchKernelGGL(len_mask_kernel, dim3((B * R + 255) / 256), dim3(256), 0, st, len, mask, B, R);
    LAUNCH_CHECK();
}

__global__ void convpos_pack_kernel(const float* __restrict__ h, const uint8_t* __restrict__ mask,
                                    bf16_t* __restrict__ gm_hi, bf16_t* __restrict__ gm_lo, int B, int T, int G, int cpg,
                                    int pad, int gstride) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int TP = T + 2 * pad;
    long total = (long)B * G * TP * gstride;
    if (i >= total) return;
    int c = (int)(i % gstride);
    long r = i / gstride;
    int tp = (int)(r % TP);
    int zz = (int)(r / TP);
    int b = zz / G, g = zz % G;
    int t = tp - pad;
    float v = 0.f;
    if (t >= 0 && t < T && c < cpg && mask[b * T + t]) v = h[((long)b * T + t) * (G * cpg) + g * cpg + c];
    store_act1(gm_hi, gm_lo, i, v);
}
hipError_t launch_convpos_pack(const float* h, const uint8_t* mask, bf16_t* gm_hi, bf16_t* gm_lo, int B, int T, int G,
                               int cpg, int pad, int gstride, hipStream_t st) {
    long total = (long)B * G * (T + 2 * pad) * gstride;
    hipLaunchKernelGGL(convpos_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, h, mask, gm_hi, gm_lo, B, T,
                       G, cpg, pad, gstride);
    LAUNCH_CHECK();
}

__global__ void axpby_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ y,
                             float a, float b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}
hipError_t launch_axpby(float* out, const float* x, const float* y, float a, float b, long n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, x, y, a, b, n);
    LAUNCH_CHECK();
}

__global__ void ode_step_kernel(float* __restrict__ xt, const float* __restrict__ v, float* __restrict__ x0_out,
                                float a, float s, float a2, float s2, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xt[i], vv = v[i];
    float x0 = a * x - s * vv;
    float ep = s * x + a * vv;
    x0_out[i] = x0;
    xt[i] = a2 * x0 + s2 * ep;
}
hipError_t launch_ode_step(float* xt, const float* v, float* x0_out, float a, float s, float a2, float s2, long n,
                           hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ode_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xt, v, x0_out, a, s, a2, s2, n);
    LAUNCH_CHECK();
}

__global__ void cfg_combine_kernel(const float* __restrict__ v3, float* __restrict__ v, float st_, float ss, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float vc = v3[i], vt = v3[n + i], vs = v3[2 * n + i];
    v[i] = vc + st_ * (vc - vt) + ss * (vc - vs);
}
hipError_t launch_cfg_combine(const float* v3, float* v, float s_text, float s_spk, long n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v3, v, s_text, s_spk, n);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__global__ void randn_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t stream) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // quad index
    if (q * 4 >= n) return;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float u1 = ((float)(c[2 * p] >> 8) + 0.5f) * 5.9604644775390625e-8f;      // (0,1)
        float u2 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
        float r = sqrtf(-2.0f * logf(u1));
        float th = 6.283185307179586f * u2;
        z[2 * p] = r * cosf(th);
        z[2 * p + 1] = r * sinf(th);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < n) out[q * 4 + j] = z[j];
}
hipError_t launch_randn(float* out, long n, uint64_t seed, uint64_t stream, hipStream_t st) {
    if (n == 0) return hipSuccess;
    long quads = (n + 3) / 4;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, out, n, seed, stream);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// synthetic weights + packing
// ------------------------------------------------------------------------------------------
__global__ void synth_kernel(float* __restrict__ out, long n, uint64_t key, float mean, float hr) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = key + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    // the recipe is "multiply, round, then add, round" (smalltts_amd/weights.py): HIP's __fmul_rn / __fadd_rn are plain operators
    // that hipcc contracts into one fma — 1 ulp off numpy wherever mean != 0 (codec layer scales: found by the converted-weight
    // test) — so contraction is switched off for this block
    {
#pragma clang fp contract(off)
        float u = (float)(uint32_t)(z >> 40);
        float s = u * 1.1920928955078125e-7f;
        s = s - 1.0f;
        float p = hr * s;
        out[i] = mean + p;
    }
}
hipError_t launch_synth(float* out, long n, uint64_t key, float mean, float half_range, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, key, mean, half_range);
    LAUNCH_CHECK();
}

__global__ void split_rows_kernel(const float* __restrict__ src, long src_ld, bf16_t* __restrict__ hi,
                                  bf16_t* __restrict__ lo, long dst_ld, int rows, int cols,
                                  const int* __restrict__ perm, bf16_t* __restrict__ h16) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    int sr = perm ? perm[r] : r;
    float v = sr >= 0 ? src[(long)sr * src_ld + c] : 0.f;
    if (hi) store_act1(hi, lo, (long)r * dst_ld + c, v);
    if (h16) store_act1(h16, SM_F16_TAG, (long)r * dst_ld + c, v);
}
hipError_t launch_split_rows(const float* src, long src_ld, bf16_t* hi, bf16_t* lo, long dst_ld, int rows, int cols,
                             const int* perm, hipStream_t st, bf16_t* h16) {
    long n = (long)rows * cols;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, src_ld, hi, lo, dst_ld,
                       rows, cols, perm, h16);
    LAUNCH_CHECK();
}

// l16 = fp16(src - float(h16)): the low half of an fp16 hi + lo weight pair (PREC_F16X2); h16 holds fp16(src) already
__global__ void f16_residual_kernel(const float* __restrict__ src, long src_ld, const bf16_t* __restrict__ h16, bf16_t* __restrict__ l16,
                                    long dst_ld, int rows, int cols) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    const long o = (long)r * dst_ld + c;
    const float hi = (float)reinterpret_cast<const half_t*>(h16)[o];
    reinterpret_cast<half_t*>(l16)[o] = (half_t)(src[(long)r * src_ld + c] - hi);
}
hipError_t launch_f16_residual(const float* src, long src_ld, const bf16_t* h16, bf16_t* l16, long dst_ld, int rows, int cols, hipStream_t st) {
    const long n = (long)rows * cols;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(f16_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, src_ld, h16, l16, dst_ld, rows, cols);
    LAUNCH_CHECK();
}

// fp16 range certificate of a fused codec FFN block (Engine::certify_codec_ffn).  The block's input is RMS-normalised,
// n = x / rms(x) o g with ||x / rms(x)||_2 <= sqrt(C), so for every hidden unit j and EVERY input
//     |gelu(W1_j . n + b1_j)| <= |W1_j . n + b1_j| <= sqrt(C) ||W1_j o g||_2 + |b1_j|        (Cauchy-Schwarz)
// out[0] = max over j of that bound, out[1] = sqrt(C) max_c |g_c| (bound of the normalised input itself).  Non-negative floats
// order like their bit patterns, so the maxima are integer atomics.  One wave per hidden unit.
__global__ __launch_bounds__(256) void ffn_range_bound_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ g, int F, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    float ss = 0.f, gm = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float gv = g ? g[c] : 1.f;
        const float v = w1[(long)j * C + c] * gv;
        ss = fmaf(v, v, ss);
        gm = fmaxf(gm, fabsf(gv));
    }
    ss = wave_sum(ss);
    gm = wave_max(gm);
    if (lane == 0) {
        const float rc = sqrtf((float)C);
        float bound = rc * sqrtf(ss) * 1.0001f + (b1 ? fabsf(b1[j]) : 0.f);   // (1.0001: rounding of this very sum)
        if (!(bound == bound)) bound = INFINITY;                              // NaN weights certify nothing
        atomicMax(reinterpret_cast<int*>(out), __float_as_int(bound));
        if (j == 0) atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(rc * gm));
    }
}
hipError_t launch_ffn_range_bound(const float* w1, const float* b1, const float* g, int F, int C, float* out2, hipStream_t st) {
    hipLaunchKernelGGL(ffn_range_bound_kernel, dim3((F + 3) / 4), dim3(256), 0, st, w1, b1, g, F, C, out2);
    LAUNCH_CHECK();
}

__global__ void fill_kernel(float* __restrict__ p, float v, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill(float* p, float v, long n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
    LAUNCH_CHECK();
}

__global__ void copy_strided_kernel(const float* __restrict__ src, long sld, float* __restrict__ dst, long dld,
                                    int rows, int cols) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    dst[(long)r * dld + c] = src[(long)r * sld + c];
}
hipError_t launch_copy_strided(const float* src, long sld, float* dst, long dld, int rows, int cols, hipStream_t st) {
    long n = (long)rows * cols;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, sld, dst, dld, rows, cols);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// codec element-wise kernels on channels-last padded images x[b][pad + t][c]
// ------------------------------------------------------------------------------------------
// causal depthwise conv (weights packed [K][C]) + layer-scale residual; one thread per (b, t, 4 channels)
__global__ __launch_bounds__(256) void dwconv_resid_kernel(float* __restrict__ x, const float* __restrict__ nrm,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, int B, int T, int C4,
                                                           int K, int pad) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * T * C4;
    if (i >= total) return;
    int c = (int)(i % C4);
    long r = i / C4;
    int t = (int)(r % T), b = (int)(r / T);
    const long frame0 = (long)b * (pad + T) + pad + t - (K - 1);
    const float4* n4 = reinterpret_cast<const float4*>(nrm);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4 acc = reinterpret_cast<const float4*>(bias)[c];
    for (int k = 0; k < K; ++k) {
        float4 nv = n4[(frame0 + k) * C4 + c];
        float4 wv = w4[(long)k * C4 + c];
        acc.x += wv.x * nv.x; acc.y += wv.y * nv.y; acc.z += wv.z * nv.z; acc.w += wv.w * nv.w;
    }
    float4 g = reinterpret_cast<const float4*>(gamma)[c];
    float4* xp = reinterpret_cast<float4*>(x) + ((long)b * (pad + T) + pad + t) * C4 + c;
    float4 xv = *xp;
    xv.x += g.x * acc.x; xv.y += g.y * acc.y; xv.z += g.z * acc.z; xv.w += g.w * acc.w;
    *xp = xv;
}
hipError_t launch_dwconv_resid(float* x, const float* n, const float* w, const float* bias, const float* gamma,
                               int B, int T, int C, int K, int pad, hipStream_t st) {
    if (C % 4 || pad < K - 1) return hipErrorInvalidValue;
    long total = (long)B * T * (C / 4);
    if (total == 0) return hipSuccess;
    ProfScope ps(st, "dwconv_resid", 2.0 * B * T * C * (K + 1), 12.0 * B * T * C);
    hipLaunchKernelGGL(dwconv_resid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, n, w, bias, gamma, B,
                       T, C / 4, K, pad);
    LAUNCH_CHECK();
}

// dwconv_resid followed by the FFN's RMSNorm of the updated row in one pass (wide codec stages): one wave per frame, the row
// stays in registers for the norm.  Same arithmetic as dwconv_resid_kernel + rmsnorm_kernel.
template <int LPR, int NV4>  // LPR lanes per frame (64: one wave, 256: the whole workgroup — few rows of many channels)
__global__ __launch_bounds__(256) void dwconv_resid_rms_kernel(float* __restrict__ x, const float* __restrict__ nrm,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ gamma, int B, int T, int C4, int K,
                                                               int pad, float eps, const float* __restrict__ nw,
                                                               bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo, RowMap ymap) {
    constexpr int RPB = 256 / LPR;
    __shared__ float red[4];
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < (long)B * T;
    const int t = live ? (int)(row % T) : 0, b = live ? (int)(row / T) : 0;
    const long frame0 = (long)b * (pad + T) + pad + t - (K - 1);
    const float4* n4 = reinterpret_cast<const float4*>(nrm);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4* xr = reinterpret_cast<float4*>(x) + ((long)b * (pad + T) + pad + t) * C4;
    float4 v[NV4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = sub + LPR * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && c < C4) {
            float4 nv[7];  // all taps of this channel group in flight together (K <= 7, checked by the launcher)
#pragma unroll
            for (int k = 0; k < 7; ++k) nv[k] = k < K ? n4[(frame0 + k) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 acc = reinterpret_cast<const float4*>(bias)[c];
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < K) {
                    const float4 wv = w4[(long)k * C4 + c];
                    acc.x += wv.x * nv[k].x; acc.y += wv.y * nv[k].y; acc.z += wv.z * nv[k].z; acc.w += wv.w * nv[k].w;
                }
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            float4 xv = xr[c];
            xv.x += g.x * acc.x; xv.y += g.y * acc.y; xv.z += g.z * acc.z; xv.w += g.w * acc.w;
            v[i] = xv;
        }
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    // the updated rows are stored after ALL channel groups' loads: a store between two groups' loads makes the second group's first
    // use an s_waitcnt vmcnt(0) (loads and stores share the counter), i.e. one full memory round trip per group instead of one per row
    const float4* nw4 = reinterpret_cast<const float4*>(nw);
    float4 gw[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = sub + LPR * i;
        gw[i] = (live && c < C4) ? nw4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) asm volatile("" : "+v"(gw[i].x), "+v"(gw[i].y), "+v"(gw[i].z), "+v"(gw[i].w));
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = sub + LPR * i;
        if (live && c < C4) xr[c] = v[i];
    }
    if (LPR == 64) {
        ss = group_sum<64>(ss);
    } else {  // fixed-order sum of the four waves' partials
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
    }
    const float rstd = 1.0f / sqrtf(ss / (float)(C4 * 4) + eps);
    if (!live) return;
    const long yo = ymap.at((int)row);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = sub + LPR * i;
        if (c < C4) {
            const float4 g = gw[i];
            store_split4(yhi, ylo, yo + c * 4, make_float4(v[i].x * rstd * g.x, v[i].y * rstd * g.y, v[i].z * rstd * g.z, v[i].w * rstd * g.w));
        }
    }
}
hipError_t launch_dwconv_resid_rms(float* x, const float* n, const float* w, const float* bias, const float* gamma, int B, int T,
                                   int C, int K, int pad, float eps, const float* norm_w, bf16_t* yhi, bf16_t* ylo, RowMap ymap,
                                   hipStream_t st) {
    const int c4 = C / 4;
    if (C % 4 || pad < K - 1 || K > 7 || c4 > 512 || !yhi) return hipErrorInvalidValue;
    const long rows = (long)B * T;
    if (rows == 0) return hipSuccess;
    ProfScope ps(st, "dwconv_resid_rms", 2.0 * B * T * C * (K + 3), 16.0 * B * T * C);
#define DWR_GO(LPR, NV) hipLaunchKernelGGL((dwconv_resid_rms_kernel<LPR, NV>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), dim3(256), 0, st, x, n, w, bias, gamma, B, T, c4, K, pad, eps, norm_w, yhi, ylo, ymap)
    if (rows < 4096 && c4 > 128) {  // few frames of many channels: a whole workgroup per frame
        if (c4 <= 256) DWR_GO(256, 1);
        else DWR_GO(256, 2);
    } else if (c4 <= 64) DWR_GO(64, 1);
    else if (c4 <= 128) DWR_GO(64, 2);
    else if (c4 <= 256) DWR_GO(64, 4);
    else DWR_GO(64, 8);
#undef DWR_GO
    LAUNCH_CHECK();
}

// head conv to 1 channel: weights packed [K][C]; 8 lanes cooperate on one output sample
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float bias, float* __restrict__ audio, int B, int T, int C,
                                                        int K, int pad) {
    const int sub = threadIdx.x & 7;
    long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool live = s < (long)B * T;
    int t = live ? (int)(s % T) : 0, b = live ? (int)(s / T) : 0;
    const float* xr = x + ((long)b * (pad + T) + pad + t - (K - 1)) * C;  // K*C contiguous floats
    const int n4 = (K * C) >> 2;
    float acc = 0.f;
    for (int i = sub; i < n4; i += 8) {
        float4 xv = reinterpret_cast<const float4*>(xr)[i];
        float4 wv = reinterpret_cast<const float4*>(w)[i];
        acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
    }
    acc = group_sum<8>(acc);
    if (live && sub == 0) audio[s] = acc + bias;
}
// C = 32 (the full-rate stage of the default spec): a workgroup stages 256 + K - 1 frames in LDS once (rows padded to 33 floats:
// lane t reads frame t + k, channel c without bank conflicts) and every thread then produces one sample from LDS, the K x 32
// weights coming from LDS as broadcasts — each frame is read from memory once instead of K times through L1.
__global__ __launch_bounds__(256) void head_conv32_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                          float* __restrict__ audio, int T, int K, int pad, int tiles_per_b) {
    constexpr int C = 32, TT = 256;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* xs = hsm;                        // [(TT + K - 1)][33]
    float* ws = hsm + (TT + 6) * 33 + 32;   // [K][32]  (K <= 7)
    const int b = blockIdx.x / tiles_per_b, t0 = (blockIdx.x % tiles_per_b) * TT;
    const int H = K - 1;
    const int nfr = (T - t0 < TT ? T - t0 : TT) + H;
    const float* x0 = x + ((long)b * (pad + T) + pad + t0 - H) * C;   // first halo frame (inside the zero pad for t0 = 0)
    for (int i = threadIdx.x; i < nfr * (C / 4); i += 256) {
        const int f = i / (C / 4), c4 = i % (C / 4);
        const float4 v = reinterpret_cast<const float4*>(x0 + (long)f * C)[c4];
        float* d = xs + f * 33 + c4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < K * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int t = threadIdx.x;
    if (t0 + t >= T) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const float* xr = xs + (t + k) * 33;
        const float* wr = ws + k * C;
#pragma unroll
        for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wr[c], acc);
    }
    audio[(long)b * T + t0 + t] = acc + bias;
}

hipError_t launch_head_conv(const float* x, const float* w, float bias, float* audio, int B, int T, int C, int K,
                            int pad, hipStream_t st) {
    if (C % 4 || pad < K - 1) return hipErrorInvalidValue;
    long threads = (long)B * T * 8;
    if (threads == 0) return hipSuccess;
    ProfScope ps(st, "head_conv", 2.0 * B * T * C * K, 4.0 * B * T * (C + 1));
    if (C == 32 && K <= 7 && T >= 256) {
        const int tiles = (T + 255) / 256;
        const size_t lds = ((size_t)(256 + 6) * 33 + 32 + 7 * 32) * sizeof(float);
        hipLaunchKernelGGL(head_conv32_kernel, dim3((unsigned)(B * tiles)), dim3(256), lds, st, x, w, bias, audio, T, K, pad, tiles);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(head_conv_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, w, bias, audio, B, T, C,
                       K, pad);
    LAUNCH_CHECK();
}

// encoder stem: 1 -> C channels, causal; weights [C][K]
__global__ void stem_conv1_kernel(const float* __restrict__ audio, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ x, int B, int T, int C, int K,
                                  int pad) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * C) return;
    int c = (int)(i % C);
    long r = i / C;
    int t = (int)(r % T), b = (int)(r / T);
    float acc = bias[c];
    for (int k = 0; k < K; ++k) {
        int tt = t - (K - 1) + k;
        if (tt >= 0) acc += w[c * K + k] * audio[(long)b * T + tt];
    }
    x[((long)b * (pad + T) + pad + t) * C + c] = acc;
}
hipError_t launch_stem_conv1(const float* audio, const float* w, const float* bias, float* x, int B, int T, int C,
                             int K, int pad, hipStream_t st) {
    long n = (long)B * T * C;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(stem_conv1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, audio, w, bias, x, B, T, C, K, pad);
    LAUNCH_CHECK();
}

__global__ void zero_pad_frames_kernel(float* __restrict__ x, int B, int T, int C, int pad) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)pad * C;
    if (i >= (long)B * per) return;
    int b = (int)(i / per);
    long o = i % per;
    x[(long)b * (pad + T) * C + o] = 0.f;
}
// the same pads of up to three images of one geometry in one launch (blockIdx.y picks the image)
__global__ void zero_pad_frames3_kernel(float* __restrict__ x0, float* __restrict__ x1, float* __restrict__ x2, int B, int T, int C, int pad) {
    float* x = blockIdx.y == 0 ? x0 : blockIdx.y == 1 ? x1 : x2;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)pad * C;
    if (i >= (long)B * per) return;
    int b = (int)(i / per);
    long o = i % per;
    x[(long)b * (pad + T) * C + o] = 0.f;
}
hipError_t launch_zero_pad_frames3(float* x0, float* x1, float* x2, int B, int T, int C, int pad, hipStream_t st) {
    long n = (long)B * pad * C;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_pad_frames3_kernel, dim3((unsigned)((n + 255) / 256), x2 ? 3 : 2), dim3(256), 0, st, x0, x1, x2, B, T, C, pad);
    LAUNCH_CHECK();
}
hipError_t launch_zero_pad_frames(float* x, int B, int T, int C, int pad, hipStream_t st) {
    long n = (long)B * pad * C;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_pad_frames_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, B, T, C, pad);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// weight re-layout gather + rope tables
// ------------------------------------------------------------------------------------------
__global__ void gather_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K, GatherSpec g) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    int n = (int)(i / K), k = (int)(i % K);
    int n1 = n / g.n0dim, n0 = n % g.n0dim, k1 = k / g.k0dim, k0 = k % g.k0dim;
    float v = 0.f;
    if (k0 < g.k0valid) v = src[g.base + n1 * g.sn1 + n0 * g.sn0 + k1 * g.sk1 + k0 * g.sk0];
    dst[i] = v;
}
hipError_t launch_gather_pack(const float* src, float* dst, int N, int K, GatherSpec g, hipStream_t st) {
    long n = (long)N * K;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, N, K, g);
    LAUNCH_CHECK();
}

__global__ void rope_table_kernel(float* __restrict__ tab, int npos, int dim) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npos * dim) return;
    int pos = i / dim, d = i % dim;
    float inv = 1.0f / powf(10000.0f, (float)(d & ~1) / (float)dim);
    tab[i] = (float)pos * inv;
}
hipError_t launch_rope_table(float* tab, int npos, int dim, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_kernel, dim3((npos * dim + 255) / 256), dim3(256), 0, st, tab, npos, dim);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// modulation-table post-pass and fp32 -> split conversion
// ------------------------------------------------------------------------------------------
__global__ void tanh_gates_kernel(float* __restrict__ mod, int rows, long ld, int n_blocks, int per_block, int hidden) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per_row = (long)n_blocks * 2 * hidden;
    if (i >= rows * per_row) return;
    int r = (int)(i / per_row);
    long j = i % per_row;
    int blk = (int)(j / (2 * hidden)), w = (int)(j % (2 * hidden));
    int col = blk * per_block + (w < hidden ? 2 * hidden + w : 5 * hidden + (w - hidden));  // gate_msa | gate_mlp
    float* p = mod + (long)r * ld + col;
    *p = tanhf(*p);
}
hipError_t launch_tanh_gates(float* mod, int rows, long ld, int n_blocks, int per_block, int hidden, hipStream_t st) {
    long n = (long)rows * n_blocks * 2 * hidden;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tanh_gates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mod, rows, ld, n_blocks, per_block, hidden);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// LN-fold tables: W shift and W (1 + scale) for every (step, block, site) in ONE launch (kernels.hpp FoldSites).
// A workgroup takes 64 weight rows of one site and FV_STEPS steps: the 2 FV_STEPS vectors sit in LDS as fp32, four lanes share a
// row (16-byte pieces of it, interleaved), every weight element is read once per step chunk.  HBM-bound on the weights
// (205 MB at fp16 for the 24 sites), fixed summation order.
// ------------------------------------------------------------------------------------------
static constexpr int FV_STEPS = 4, FV_ROWS = 64, FV_K = 960;
__global__ __launch_bounds__(256) void fold_vectors_kernel(FoldSites fs, const float* __restrict__ mod, long mod_ld, int rows,
                                                           float* __restrict__ tab) {
    __shared__ __attribute__((aligned(16))) float vec[2 * FV_STEPS][FV_K];
    // which site: blockIdx.x counts 64-row groups over the sites in order
    int si = 0, g0 = blockIdx.x;
    for (; si < fs.n; ++si) {
        const int ng = (fs.s[si].N + FV_ROWS - 1) / FV_ROWS;
        if (g0 < ng) break;
        g0 -= ng;
    }
    if (si >= fs.n) return;
    const FoldSite S = fs.s[si];
    const int step0 = blockIdx.y * FV_STEPS;
    for (int i = threadIdx.x; i < 2 * FV_STEPS * FV_K; i += 256) {
        const int v = i / FV_K, k = i - v * FV_K, st = step0 + (v >> 1);
        float x = 0.f;
        if (st < rows) x = (v & 1) ? 1.0f + mod[(long)st * mod_ld + S.scale_off + k] : mod[(long)st * mod_ld + S.shift_off + k];
        vec[v][k] = x;
    }
    __syncthreads();
    const int q = threadIdx.x & 3, n = g0 * FV_ROWS + (threadIdx.x >> 2);
    const int nn = n < S.N ? n : S.N - 1;
    const uint4* wr = reinterpret_cast<const uint4*>(S.w + (long)nn * FV_K);
    const uint4* wl = S.wlo ? reinterpret_cast<const uint4*>(S.wlo + (long)nn * FV_K) : nullptr;
    float acc[2 * FV_STEPS];
#pragma unroll
    for (int v = 0; v < 2 * FV_STEPS; ++v) acc[v] = 0.f;
#pragma unroll 2
    for (int i = 0; i < FV_K / 32; ++i) {
        const int c = q + 4 * i;   // 16-byte piece of the row: elements [8 c, 8 c + 8)
        const uint4 u = wr[c];
        float w[8];
        const unsigned uu[4] = {u.x, u.y, u.z, u.w};
        if (S.fmt == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2_t h = __builtin_bit_cast(half2_t, uu[e]);
                w[2 * e] = (float)h[0]; w[2 * e + 1] = (float)h[1];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { w[2 * e] = __uint_as_float(uu[e] << 16); w[2 * e + 1] = __uint_as_float(uu[e] & 0xffff0000u); }
            if (wl) {
                const uint4 l = wl[c];
                const unsigned ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { w[2 * e] += __uint_as_float(ll[e] << 16); w[2 * e + 1] += __uint_as_float(ll[e] & 0xffff0000u); }
            }
        }
#pragma unroll
        for (int v = 0; v < 2 * FV_STEPS; ++v) {
            const float4 a = *reinterpret_cast<const float4*>(&vec[v][8 * c]);
            const float4 b = *reinterpret_cast<const float4*>(&vec[v][8 * c + 4]);
            float t = acc[v];
            t = fmaf(w[0], a.x, t); t = fmaf(w[1], a.y, t); t = fmaf(w[2], a.z, t); t = fmaf(w[3], a.w, t);
            t = fmaf(w[4], b.x, t); t = fmaf(w[5], b.y, t); t = fmaf(w[6], b.z, t); t = fmaf(w[7], b.w, t);
            acc[v] = t;
        }
    }
#pragma unroll
    for (int v = 0; v < 2 * FV_STEPS; ++v) {
        acc[v] += __shfl_xor(acc[v], 1, 64);
        acc[v] += __shfl_xor(acc[v], 2, 64);
    }
    if (q == 0 && n < S.N) {
#pragma unroll
        for (int v = 0; v < 2 * FV_STEPS; ++v) {
            const int st = step0 + (v >> 1);
            if (st < rows) tab[((long)st * 2 + (v & 1)) * fs.NF + S.out_off + n] = acc[v];
        }
    }
}
hipError_t launch_fold_vectors(const FoldSites& sites, const float* mod, long mod_ld, int rows, float* tab, hipStream_t st) {
    if (rows <= 0 || sites.n <= 0) return hipSuccess;
    long groups = 0, wbytes = 0;
    for (int i = 0; i < sites.n; ++i) {
        groups += (sites.s[i].N + FV_ROWS - 1) / FV_ROWS;
        wbytes += (long)sites.s[i].N * FV_K * (sites.s[i].wlo ? 4 : 2);
    }
    const int chunks = (rows + FV_STEPS - 1) / FV_STEPS;
    ProfScope ps(st, "fold_vectors", 4.0 * sites.NF * FV_K * rows, (double)wbytes * chunks + 8.0 * sites.NF * rows, 0.0);
    hipLaunchKernelGGL(fold_vectors_kernel, dim3((unsigned)groups, (unsigned)chunks), dim3(256), 0, st, sites, mod, mod_ld, rows, tab);
    LAUNCH_CHECK();
}

__global__ void to_split_kernel(const float* __restrict__ x, RowMap xmap, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                RowMap omap, int M, int C4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * C4) return;
    int m = (int)(i / C4), c = (int)(i % C4);
    float4 v = reinterpret_cast<const float4*>(x + xmap.at(m))[c];
    store_split4(hi, lo, omap.at(m) + c * 4, v);
}
hipError_t launch_to_split(const float* x, RowMap xmap, bf16_t* hi, bf16_t* lo, RowMap omap, int M, int C, hipStream_t st) {
    long n = (long)M * (C / 4);
    if (n == 0) return hipSuccess;
    ProfScope ps(st, "to_split", 0, 8.0 * M * C);
    hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xmap, hi, lo, omap, M, C / 4);
    LAUNCH_CHECK();
}

__global__ void rope_cossin_kernel(const float* __restrict__ ang, float* __restrict__ c, float* __restrict__ s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = ang[i];
    c[i] = cosf(a);
    s[i] = sinf(a);
}
hipError_t launch_rope_cossin(const float* ang, float* c, float* s, int n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rope_cossin_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ang, c, s, n);
    LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void splitk_resid_kernel(const float* __restrict__ part, int S, float* __restrict__ x,
                                                           const float* __restrict__ bias, const float* __restrict__ gate,
                                                           long gld, int grow0, int grstride, int rpb,
                                                           const uint8_t* __restrict__ rowmask, int M, int N4, RowMap xmap,
                                                           int overwrite) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N4) return;
    const int m = (int)(i / N4), c = (int)(i % N4);
    if (rowmask && !rowmask[m]) return;
    const long MN4 = (long)M * N4;
    float4 acc = reinterpret_cast<const float4*>(part)[i];
    for (int s = 1; s < S; ++s) {
        const float4 p = reinterpret_cast<const float4*>(part)[s * MN4 + i];
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    if (bias) {
        const float4 b = reinterpret_cast<const float4*>(bias)[c];
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gate) g = reinterpret_cast<const float4*>(gate + (long)(grow0 + (m / rpb) * grstride) * gld)[c];
    float4* xr = reinterpret_cast<float4*>(x + xmap.at(m)) + c;  // residual image row m (plain [M][N] or a padded codec image)
    float4 xv = overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *xr;   // overwrite: x = gate * (sum + bias), closes a split-K product that has no residual
    xv.x += g.x * acc.x; xv.y += g.y * acc.y; xv.z += g.z * acc.z; xv.w += g.w * acc.w;
    *xr = xv;
}
// splitk_resid followed by the NEXT AdaLN (ln_modulate) in one pass over the row: one wave per row, the updated
// residual row stays in registers for the LayerNorm.  Same arithmetic order as the two separate kernels.
template <int NV4, bool RMS>
__global__ __launch_bounds__(256) void splitk_resid_ln_kernel(const float* __restrict__ part, int S, float* __restrict__ x,
                                                              const float* __restrict__ bias, const float* __restrict__ gate,
                                                              long gld, int grow0, int grstride, int rpb,
                                                              const uint8_t* __restrict__ rowmask, int M, int C, float eps,
                                                              const float* __restrict__ shift, const float* __restrict__ scale,
                                                              bf16_t* __restrict__ yhi, bf16_t* __restrict__ ylo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int C4 = C >> 2;
    const long MN4 = (long)M * C4;
    const long mrow = (long)(grow0 + (row / rpb) * grstride) * gld;
    float4* xr = reinterpret_cast<float4*>(x + (long)row * C);
    // RMS mode (encoders): `shift` is the RMSNorm weight [C], `scale` is unused; otherwise AdaLN rows of the modulation table
    const float4* sh4 = reinterpret_cast<const float4*>(RMS ? shift : shift + mrow);
    const float4* sc4 = reinterpret_cast<const float4*>(RMS ? shift : scale + mrow);
    // every load of the row is issued up front and unconditionally (the row-mask byte only selects afterwards), so
    // the kernel is one memory round trip + two wave reductions
    const uint8_t mk = rowmask ? rowmask[row] : (uint8_t)1;
    float4 v[NV4], sh[NV4], sc[NV4], acc[NV4], g[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = lane + 64 * i;
        const int cc = c < C4 ? c : 0;
        v[i] = xr[cc];
        sh[i] = sh4[cc];
        sc[i] = sc4[cc];
        const float4* pr = reinterpret_cast<const float4*>(part) + (long)row * C4 + cc;
        acc[i] = pr[0];
        for (int s = 1; s < S; ++s) {
            const float4 p = pr[s * MN4];
            acc[i].x += p.x; acc[i].y += p.y; acc[i].z += p.z; acc[i].w += p.w;
        }
        if (bias) {
            const float4 b = reinterpret_cast<const float4*>(bias)[cc];
            acc[i].x += b.x; acc[i].y += b.y; acc[i].z += b.z; acc[i].w += b.w;
        }
        g[i] = gate ? reinterpret_cast<const float4*>(gate + mrow)[cc] : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const bool live = mk != 0;
    // every loaded value is complete before the first store: loads and stores share vmcnt and complete out of order with respect to
    // each other, so the AdaLN rows (first used after the x stores) would otherwise cost an s_waitcnt vmcnt(0) that drains those stores
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        asm volatile("" : "+v"(sh[i].x), "+v"(sh[i].y), "+v"(sh[i].z), "+v"(sh[i].w));
        asm volatile("" : "+v"(sc[i].x), "+v"(sc[i].y), "+v"(sc[i].z), "+v"(sc[i].w));
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = lane + 64 * i;
        if (live) {
            v[i].x += g[i].x * acc[i].x; v[i].y += g[i].y * acc[i].y; v[i].z += g[i].z * acc[i].z; v[i].w += g[i].w * acc[i].w;
            if (c < C4) xr[c] = v[i];
        }
        if (c >= C4) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (RMS) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV4; ++i) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = lane + 64 * i;
            if (c < C4) {
                const float4 o = make_float4(v[i].x * rstd * sh[i].x, v[i].y * rstd * sh[i].y, v[i].z * rstd * sh[i].z,
                                             v[i].w * rstd * sh[i].w);
                store_split4(yhi, ylo, (long)row * C + c * 4, o);
            }
        }
        return;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        if (lane + 64 * i < C4) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 o = make_float4((v[i].x - mean) * rstd * (1.0f + sc[i].x) + sh[i].x,
                                         (v[i].y - mean) * rstd * (1.0f + sc[i].y) + sh[i].y,
                                         (v[i].z - mean) * rstd * (1.0f + sc[i].z) + sh[i].z,
                                         (v[i].w - mean) * rstd * (1.0f + sc[i].w) + sh[i].w);
            store_split4(yhi, ylo, (long)row * C + c * 4, o);
        }
    }
}

hipError_t launch_splitk_resid_ln(const float* part, int S, float* x, const float* bias, const float* gate, long gld,
                                  int grow0, int grstride, int rows_per_batch, const uint8_t* rowmask, int M, int N, float eps,
                                  const float* shift, const float* scale, bf16_t* yhi, bf16_t* ylo, hipStream_t st, bool rms) {
    if (N > 1024 || N % 4 || gld % 4) return hipErrorInvalidValue;
    ProfScope ps(st, rms ? "splitk_resid_rms" : "splitk_resid_ln", 1.0 * M * N * (S + 10), 4.0 * M * N * (S + 3));
    const dim3 grid((M + 3) / 4), block(256);
    if (rms)
        hipLaunchKernelGGL((splitk_resid_ln_kernel<4, true>), grid, block, 0, st, part, S, x, bias, gate, gld, grow0, grstride,
                           rows_per_batch, rowmask, M, N, eps, shift, scale, yhi, ylo);
    else
        hipLaunchKernelGGL((splitk_resid_ln_kernel<4, false>), grid, block, 0, st, part, S, x, bias, gate, gld, grow0, grstride,
                           rows_per_batch, rowmask, M, N, eps, shift, scale, yhi, ylo);
    LAUNCH_CHECK();
}

hipError_t launch_splitk_resid(const float* part, int S, float* x, const float* bias, const float* gate, long gld,
                               int grow0, int grstride, int rows_per_batch, const uint8_t* rowmask, int M, int N,
                               hipStream_t st, const RowMap* xmap, bool overwrite) {
    const RowMap xm = xmap ? *xmap : rowmap_plain(N);
    if (N % 4 || gld % 4 || xm.ld % 4 || xm.off % 4 || xm.bstride % 4 || rows_per_batch <= 0) return hipErrorInvalidValue;
    long n = (long)M * (N / 4);
    if (n == 0) return hipSuccess;
    ProfScope ps(st, "splitk_resid", 1.0 * M * N * (S + 2), 4.0 * M * N * (S + 2));
    hipLaunchKernelGGL(splitk_resid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, S, x, bias, gate, gld,
                       grow0, grstride, rows_per_batch, rowmask, M, N / 4, xm, overwrite ? 1 : 0);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// fused RMSNorm + causal depthwise conv + LayerScale residual (codec mixer), C <= 256, out of place
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mixer_fused_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                          const float* __restrict__ norm_w, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          int T, int C, int K, int pad, float eps, int TT, int tiles_per_b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C4 = C >> 2, H = K - 1;
    float* xs = sm;                         // [(TT + H)][C]
    float* rs = sm + (size_t)(TT + H) * C;  // [(TT + H)] rstd per frame
    const int b = blockIdx.x / tiles_per_b, t0 = (blockIdx.x % tiles_per_b) * TT;
    const int nfr = (T - t0 < TT ? T - t0 : TT) + H;             // frames staged (incl. halo)
    const long img0 = ((long)b * (pad + T) + pad + t0 - H) * C;  // first halo frame (inside the zero pad for t0 = 0)
    const int tid = threadIdx.x;
    // phase 1: stage frames, per-frame sum of squares (C4 lanes cooperate on one frame).  All global loads of the tile
    // are issued before the first reduction / LDS store: one memory round trip per workgroup instead of one per pass.
    const int lpr = C4, fpp = 256 / lpr;
    constexpr int NI = 10;  // (TT + H) * C4 <= 256 * NI for TT * C = 8192, H <= 6 (checked by the launcher)
    float4 stg[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = it * fpp + tid / lpr, c4 = tid % lpr;
        stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nfr) stg[it] = reinterpret_cast<const float4*>(xin + img0 + (long)f * C)[c4];
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int f = it * fpp + tid / lpr, c4 = tid % lpr;
        const float4 v = stg[it];
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        for (int o = lpr >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (f < nfr) {
            // staged NORMALISED (u = x * rstd): the seven taps of every output then cost one fma per value instead of a
            // multiply by the frame's rstd + an fma and an LDS read of that rstd (round 4); the raw value the residual needs is
            // u * rms of the centre tap (rs holds the rms, not its reciprocal)
            const float rms = sqrtf(ss / (float)C + eps), r = 1.0f / rms;
            reinterpret_cast<float4*>(xs + (size_t)f * C)[c4] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
            if (c4 == 0) rs[f] = rms;
        }
    }
    // per-thread constants: 256 % C4 == 0, so a thread always works on the same 4 channels
    const int c4 = tid % C4;
    float4 wv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wv[k] = k < K ? reinterpret_cast<const float4*>(w)[(long)k * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = reinterpret_cast<const float4*>(norm_w)[c4];
    const float4 bb = reinterpret_cast<const float4*>(bias)[c4];
    const float4 gm = reinterpret_cast<const float4*>(gamma)[c4];
    __syncthreads();
    // phase 2: the tile's output frames
    const int nout = nfr - H;
    for (int t = tid / C4; t < nout; t += fpp) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 uc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (k < K) {
                const float4 u = reinterpret_cast<const float4*>(xs + (size_t)(t + k) * C)[c4];
                acc.x = fmaf(wv[k].x, u.x, acc.x); acc.y = fmaf(wv[k].y, u.y, acc.y);
                acc.z = fmaf(wv[k].z, u.z, acc.z); acc.w = fmaf(wv[k].w, u.w, acc.w);
                if (k == H) uc = u;
            }
        }
        const float rms = rs[t + H];
        float4 xv;
        xv.x = fmaf(uc.x, rms, gm.x * (g.x * acc.x + bb.x)); xv.y = fmaf(uc.y, rms, gm.y * (g.y * acc.y + bb.y));
        xv.z = fmaf(uc.z, rms, gm.z * (g.z * acc.z + bb.z)); xv.w = fmaf(uc.w, rms, gm.w * (g.w * acc.w + bb.w));
        reinterpret_cast<float4*>(xout + img0 + (long)(t + H) * C)[c4] = xv;
    }
}
// The same arithmetic, in the same order (bit-identical results), as a STREAM along time for C = 128 / 256 (round 4): a group of C / 4
// lanes owns a segment of L consecutive frames of one utterance and walks it with the last six normalised frames in registers — the
// conv needs no LDS and no workgroup barrier, every wave is independent, and a chunk of eight frames is requested before the first is
// used.  The staged kernel above loads a tile, synchronises, computes, stores: its workgroups' phases overlap only through
// occupancy (4.4 TB/s of algorithmic bytes at C = 128).
template <int C, int U>
__global__ __launch_bounds__(256) void mixer_stream_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                           const float* __restrict__ norm_w, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           int T, int pad, float eps, int L, int segs_per_b, int nseg) {
    constexpr int C4 = C / 4, GPW = 64 / C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = lane % C4;
    const int seg = (blockIdx.x * 4 + wave) * GPW + lane / C4;
    if (seg >= nseg) return;   // (a whole group: the shuffles below stay inside a group)
    const int b = seg / segs_per_b, t0 = (seg % segs_per_b) * L;
    const int nfr = T - t0 < L ? T - t0 : L;
    const long base = ((long)b * (pad + T) + pad + t0 - 6) * C + c4 * 4;   // first halo frame (the zero pad for t0 = 0)
    float4 wv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wv[k] = reinterpret_cast<const float4*>(w)[(long)k * C4 + c4];
    const float4 g = reinterpret_cast<const float4*>(norm_w)[c4];
    const float4 bb = reinterpret_cast<const float4*>(bias)[c4];
    const float4 gm = reinterpret_cast<const float4*>(gamma)[c4];
    float4 win[6];   // u of frames f - 6 .. f - 1
#pragma unroll
    for (int k = 0; k < 6; ++k) win[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int total = nfr + 6;   // frames walked: six of halo, then the segment
#pragma unroll 1
    for (int f0 = 0; f0 < total; f0 += U) {
        float4 st[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int f = f0 + i;
            st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < total) st[i] = *reinterpret_cast<const float4*>(xin + base + (long)f * C);
        }
        float rmsv[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {   // the eight reductions are independent chains
            float ss = st[i].x * st[i].x + st[i].y * st[i].y + st[i].z * st[i].z + st[i].w * st[i].w;
#pragma unroll
            for (int o = C4 >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            rmsv[i] = sqrtf(ss / (float)C + eps);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int f = f0 + i;
            const float r = 1.0f / rmsv[i];
            const float4 u = make_float4(st[i].x * r, st[i].y * r, st[i].z * r, st[i].w * r);
            if (f >= 6 && f < total) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    acc.x = fmaf(wv[k].x, win[k].x, acc.x); acc.y = fmaf(wv[k].y, win[k].y, acc.y);
                    acc.z = fmaf(wv[k].z, win[k].z, acc.z); acc.w = fmaf(wv[k].w, win[k].w, acc.w);
                }
                acc.x = fmaf(wv[6].x, u.x, acc.x); acc.y = fmaf(wv[6].y, u.y, acc.y);
                acc.z = fmaf(wv[6].z, u.z, acc.z); acc.w = fmaf(wv[6].w, u.w, acc.w);
                float4 xv;
                xv.x = fmaf(u.x, rmsv[i], gm.x * (g.x * acc.x + bb.x)); xv.y = fmaf(u.y, rmsv[i], gm.y * (g.y * acc.y + bb.y));
                xv.z = fmaf(u.z, rmsv[i], gm.z * (g.z * acc.z + bb.z)); xv.w = fmaf(u.w, rmsv[i], gm.w * (g.w * acc.w + bb.w));
                *reinterpret_cast<float4*>(xout + base + (long)f * C) = xv;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) win[k] = win[k + 1];
            win[5] = u;
        }
    }
}

// The same mixer for the WIDE stages (C = 512 / 1024 / 2048), plus the FFN's RMSNorm of the updated rows: one pass over the image
//     x_mid = x + gamma * (norm_w * conv7(x * rstd(x)) + dw_b)            -> xout (fp32 image, out of place)
//     n2    = x_mid * rstd(x_mid) * ffn_norm_w                            -> 16-bit operand rows of the first FFN product
// instead of rmsnorm (x -> fp32 normalised image) + dwconv_resid_rms (reads both): the normalised image (one write + one read of
// the stage's rows) and a launch per block go.  A workgroup stages TT + 6 raw frames of one utterance in LDS, computes their
// rstd (one wave per frame), then every thread owns 4 * CPT channels: conv over the seven staged rows, residual, partial sums of
// squares per output frame (reduced over the waves that share a frame through LDS), and the two stores.  All global loads of the
// tile are issued before the first use; stores only after the last load (shared vmcnt, NOTEBOOK §5a).
template <int C, int TT>
__global__ __launch_bounds__(256) void mixer_wide_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                         const float* __restrict__ norm_w, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ gamma,
                                                         const float* __restrict__ ffn_norm_w, bf16_t* __restrict__ n2hi,
                                                         bf16_t* __restrict__ n2lo, int T, int pad, float eps, int tiles_per_b) {
    constexpr int C4 = C / 4, H = 6, NF = TT + H;
    constexpr int LPR = C4 < 256 ? C4 : 256;   // threads that share a frame
    constexpr int FPP = 256 / LPR;             // frames worked on in parallel
    constexpr int CPT = C4 / LPR;              // float4 channel groups per thread
    constexpr int FT = TT / FPP;               // output frames per thread
    constexpr int NI = (NF * C4 + 255) / 256;  // staging loads per thread
    constexpr int WPF = LPR / 64;              // waves that share a frame
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                    // [NF][C] raw frames
    float* rs = sm + (size_t)NF * C;   // [NF] rstd of the raw frames
    float* red = rs + NF;              // [TT][WPF] partial sums of squares of the updated frames
    const int b = blockIdx.x / tiles_per_b, t0 = (blockIdx.x % tiles_per_b) * TT;
    const int nout = T - t0 < TT ? T - t0 : TT, nfr = nout + H;
    const long img0 = ((long)b * (pad + T) + pad + t0 - H) * C;  // first halo frame (inside the zero pad for t0 = 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        float4 stg[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = it * 256 + tid, f = i / C4, c4 = i % C4;
            stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < nfr) stg[it] = reinterpret_cast<const float4*>(xin + img0 + (long)f * C)[c4];
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = it * 256 + tid, f = i / C4, c4 = i % C4;
            if (f < nfr) reinterpret_cast<float4*>(xs + (size_t)f * C)[c4] = stg[it];
        }
    }
    // per-thread constants (loaded while the tile settles)
    const int cg = tid % LPR, fsel = tid / LPR;
    float4 wv[CPT][7], g[CPT], bb[CPT], gm[CPT], w2[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c4 = cg + j * LPR;
#pragma unroll
        for (int k = 0; k < 7; ++k) wv[j][k] = reinterpret_cast<const float4*>(w)[(long)k * C4 + c4];
        g[j] = reinterpret_cast<const float4*>(norm_w)[c4];
        bb[j] = reinterpret_cast<const float4*>(bias)[c4];
        gm[j] = reinterpret_cast<const float4*>(gamma)[c4];
        w2[j] = reinterpret_cast<const float4*>(ffn_norm_w)[c4];
    }
    __syncthreads();
    for (int f = wave; f < nfr; f += 4) {   // rstd of the staged frames: one wave per frame
        float ss = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const float4 v = reinterpret_cast<const float4*>(xs + (size_t)f * C)[c4];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane == 0) rs[f] = 1.0f / sqrtf(ss / (float)C + eps);
    }
    __syncthreads();
    float4 xv[FT][CPT];
    float ssp[FT];
#pragma unroll
    for (int ti = 0; ti < FT; ++ti) {
        const int t = fsel + ti * FPP;
        ssp[ti] = 0.f;
        if (t < nout) {
            float r[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) r[k] = rs[t + k];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c4 = cg + j * LPR;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float4 u = reinterpret_cast<const float4*>(xs + (size_t)(t + k) * C)[c4];
                    acc.x += wv[j][k].x * r[k] * u.x; acc.y += wv[j][k].y * r[k] * u.y;
                    acc.z += wv[j][k].z * r[k] * u.z; acc.w += wv[j][k].w * r[k] * u.w;
                }
                float4 v = reinterpret_cast<const float4*>(xs + (size_t)(t + H) * C)[c4];
                v.x += gm[j].x * (g[j].x * acc.x + bb[j].x); v.y += gm[j].y * (g[j].y * acc.y + bb[j].y);
                v.z += gm[j].z * (g[j].z * acc.z + bb[j].z); v.w += gm[j].w * (g[j].w * acc.w + bb[j].w);
                xv[ti][j] = v;
                ssp[ti] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
    }
#pragma unroll
    for (int ti = 0; ti < FT; ++ti) {
        float ss = ssp[ti];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane == 0) red[(fsel + ti * FPP) * WPF + (wave % WPF)] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < FT; ++ti) {
        const int t = fsel + ti * FPP;
        if (t < nout) {
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < WPF; ++q) ss += red[t * WPF + q];
            const float r2 = 1.0f / sqrtf(ss / (float)C + eps);
            const long row = (long)b * T + t0 + t;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c4 = cg + j * LPR;
                const float4 v = xv[ti][j];
                reinterpret_cast<float4*>(xout + img0 + (long)(t + H) * C)[c4] = v;
                store_split4(n2hi, n2lo, row * C + c4 * 4,
                             make_float4(v.x * r2 * w2[j].x, v.y * r2 * w2[j].y, v.z * r2 * w2[j].z, v.w * r2 * w2[j].w));
            }
        }
    }
}
bool mixer_wide_ok(int C, int K) { return K == 7 && (C == 512 || C == 1024 || C == 2048); }
hipError_t launch_mixer_wide(const float* xin, float* xout, const float* norm_w, const float* w, const float* bias, const float* gamma,
                             const float* ffn_norm_w, bf16_t* n2hi, bf16_t* n2lo, int B, int T, int C, int K, int pad, float eps,
                             hipStream_t st) {
    if (!mixer_wide_ok(C, K) || pad < K - 1 || xin == xout || !n2hi) return hipErrorInvalidValue;
    if ((long)B * T == 0) return hipSuccess;
    ProfScope ps(st, "mixer_wide", 2.0 * B * T * C * (K + 7), 10.0 * B * T * C);
    // Frames per tile: SMALL tiles win — the phases of a workgroup (load, rstd, conv, reduce, store) are serial, so the CU needs
    // several workgroups to overlap them, and the halo re-reads (6 frames per tile) are L2 hits.  Measured, us per launch at
    // TT = 32 / 16 / 8 / 4 (profiles/r03z_*): C = 512 (24000 rows) 68.6 / 38.5 / 36.3 / 40.2; C = 1024 (4800 rows) - / 42.4 /
    // 27.4 / 23.8; C = 2048 (600 rows) - / - / 18.0 / 13.7 — against 64.2 / 37.8 / 23.7 for the two kernels this replaces.
#define MW_GO(CC, TT)                                                                                                              \
    do {                                                                                                                           \
        constexpr int WPF = ((CC) / 4 < 256 ? (CC) / 4 : 256) / 64;                                                                \
        const size_t lds = ((size_t)((TT) + 6) * (CC) + ((TT) + 6) + (size_t)(TT) * WPF) * sizeof(float);                          \
        auto kern = mixer_wide_kernel<CC, TT>;                                                                                     \
        static DevOnce once;                                                                                                       \
        hipError_t e = once.ensure([&] {                                                                                           \
            return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        });                                                                                                                        \
        if (e != hipSuccess) return e;                                                                                             \
        const int tiles = (T + (TT) - 1) / (TT);                                                                                   \
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(256), lds, st, xin, xout, norm_w, w, bias, gamma, ffn_norm_w,   \
                           n2hi, n2lo, T, pad, eps, tiles);                                                                        \
    } while (0)
    static const int tt512 = lab_env("SMTTS_MW_TT512") ? atoi(lab_env("SMTTS_MW_TT512")) : 8;
    static const int tt1024 = lab_env("SMTTS_MW_TT1024") ? atoi(lab_env("SMTTS_MW_TT1024")) : 4;
    static const int tt2048 = lab_env("SMTTS_MW_TT2048") ? atoi(lab_env("SMTTS_MW_TT2048")) : 4;
    if (C == 2048) { if (tt2048 == 4) MW_GO(2048, 4); else MW_GO(2048, 8); }
    else if (C == 1024) { if (tt1024 == 4) MW_GO(1024, 4); else if (tt1024 == 8) MW_GO(1024, 8); else MW_GO(1024, 16); }
    else { if (tt512 == 4) MW_GO(512, 4); else if (tt512 == 8) MW_GO(512, 8); else if (tt512 == 16) MW_GO(512, 16); else MW_GO(512, 32); }
#undef MW_GO
    LAUNCH_CHECK();
}

hipError_t launch_mixer_fused(const float* xin, float* xout, const float* norm_w, const float* w, const float* bias,
                              const float* gamma, int B, int T, int C, int K, int pad, float eps, hipStream_t st) {
    if (C % 4 || C > 256 || (256 % (C / 4)) || pad < K - 1 || xin == xout || K > 7) return hipErrorInvalidValue;
    static const int stream_on = getenv("SMTTS_MIXER_STREAM") ? atoi(getenv("SMTTS_MIXER_STREAM")) : 1;   // A/B: 0 = the staged kernel
    if (stream_on && (C == 128 || C == 256) && K == 7 && pad >= 6 && (long)B * T > 0) {
        const int L = stream_on > 1 ? stream_on : 42;   // frames per segment: 6 halo + 42 = six chunks of eight (sweep: profiles/r04t_*)
        const int segs = (T + L - 1) / L, nseg = B * segs, gpw = 64 / (C / 4);
        const unsigned grid = (unsigned)((nseg + 4 * gpw - 1) / (4 * gpw));
        ProfScope ps(st, "mixer_fused", 2.0 * B * T * C * (K + 4), 8.0 * B * T * C);
        if (C == 128)
            hipLaunchKernelGGL((mixer_stream_kernel<128, 8>), dim3(grid), dim3(256), 0, st, xin, xout, norm_w, w, bias, gamma, T, pad, eps, L, segs, nseg);
        else
            hipLaunchKernelGGL((mixer_stream_kernel<256, 8>), dim3(grid), dim3(256), 0, st, xin, xout, norm_w, w, bias, gamma, T, pad, eps, L, segs, nseg);
        LAUNCH_CHECK();
    }
    int TT = 8192 / C;
    if (TT < 8) TT = 8;
    if ((long)(TT + K - 1) * (C / 4) > 2560) return hipErrorInvalidValue;  // staging registers of the kernel (NI = 10)
    const int tiles = (T + TT - 1) / TT;
    const size_t lds = ((size_t)(TT + K - 1) * C + (TT + K - 1)) * sizeof(float);
    if ((long)B * tiles == 0) return hipSuccess;
    ProfScope ps(st, "mixer_fused", 2.0 * B * T * C * (K + 4), 8.0 * B * T * C);
    hipLaunchKernelGGL(mixer_fused_kernel, dim3((unsigned)(B * tiles)), dim3(256), lds, st, xin, xout, norm_w, w, bias, gamma,
                       T, C, K, pad, eps, TT, tiles);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------
// Device-side audio front / back end
// ------------------------------------------------------------------------------------------
// Polyphase windowed-sinc resampler (reference infer/utils.py:7-16 = torchaudio Resample): output sample
// f * up + p of channel c = sum_k xpad[c][f * down + k] * bank[p][k], where xpad is x with `width` zeros in front
// (and zeros behind).  One thread per output sample; the bank row is shared by every thread with the same phase.
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, long n_in, const float* __restrict__ bank,
                                                            int up, int down, int klen, int width, float* __restrict__ y,
                                                            long n_out, int channels) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out * channels) return;
    const int c = (int)(i / n_out);
    const long o = i % n_out;
    const long f = o / up;
    const int p = (int)(o % up);
    const float* xc = x + (long)c * n_in;
    const float* b = bank + (long)p * klen;
    const long s0 = f * down - width;  // index into x of tap 0
    float acc = 0.f;
    int k0 = s0 < 0 ? (int)(-s0) : 0;
    long k1 = n_in - s0;
    if (k1 > klen) k1 = klen;
    for (int k = k0; k < k1; ++k) acc = fmaf(xc[s0 + k], b[k], acc);
    y[i] = acc;
}
hipError_t launch_resample_poly(const float* x, long n_in, const float* bank, int up, int down, int klen, int width, float* y,
                                long n_out, int channels, hipStream_t st) {
    const long n = n_out * channels;
    if (n <= 0) return hipSuccess;
    ProfScope ps(st, "resample_poly", 2.0 * n * klen, 4.0 * (n + (double)n_in * channels));
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n_in, bank, up, down, klen,
                       width, y, n_out, channels);
    LAUNCH_CHECK();
}

// float [-1, 1] -> int16: clamp, scale by 32767, round to nearest even (reference server audio.rs:22-37; tryme.py:29 PCM_16)
__global__ void pcm16_kernel(const float* __restrict__ x, int16_t* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = fminf(fmaxf(x[i], -1.0f), 1.0f) * 32767.0f;
    y[i] = (int16_t)__float2int_rn(v);
}
hipError_t launch_pcm16(const float* x, int16_t* y, long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    LAUNCH_CHECK();
}

// t[i] = float32(np.linspace(1, 0, n))[i]: float64 arithmetic, exact end point (reference infer/onnx.py:98)
__global__ void linspace10_kernel(float* __restrict__ t, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 1.0f;
    // numpy: arange(n) * step + start, two separately rounded float64 operations (no fused multiply-add)
    if (n > 1) v = i == n - 1 ? 0.0f : (float)__dadd_rn(1.0, __dmul_rn(-1.0 / (double)(n - 1), (double)i));
    t[i] = v;
}
hipError_t launch_linspace10(float* t, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(linspace10_kernel, dim3((n + 255) / 256), dim3(256), 0, st, t, n);
    LAUNCH_CHECK();
}
