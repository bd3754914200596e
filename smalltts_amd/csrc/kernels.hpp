// Launchers for the non-GEMM kernels of the smalltts gfx950 library (definitions in kernels.hip,
// attention.hip, codec_kernels.hip).  All pointers are device pointers; all launches are async on `st`.
#pragma once
#include "common.hpp"

// y[m][c] = LN(x[m][:]; eps, no affine)[c] * (1 + scale[r][c]) + shift[r][c],  r = mod_row0 + (m / rows_per_batch) * mod_rstride
// (reference dit.py:19-25, 197-199, 35-39).  shift/scale are rows of the modulation table (ld = mod_ld).
// Output: fp32 `y`, or (when yhi != null) the split bf16 pair yhi/ylo that feeds gemm3 directly.
hipError_t launch_ln_modulate(const float* x, float* y, bf16_t* yhi, bf16_t* ylo, int M, int C, float eps,
                              const float* shift, const float* scale, long mod_ld, int mod_row0, int mod_rstride,
                              int rows_per_batch, hipStream_t st);

// y[m][c] = x[m][c] * rsqrt(mean(x[m][:]^2) + eps) * w[c]   (reference dit.py:42-53, 1-D weight).
// Rows are addressed through RowMaps so padded codec images can be normalised in place of a copy.
hipError_t launch_rmsnorm(const float* x, RowMap xmap, float* y, bf16_t* yhi, bf16_t* ylo, RowMap ymap, int M, int C,
                          float eps, const float* w, hipStream_t st);

// In-place per-head RMSNorm of the cross-K cache [L][B][H][S][dh] with weights [L][H][dh] (dit.py:83).
hipError_t launch_headnorm(float* k, int L, int B, int H, int S, int dh, float eps, const float* w, hipStream_t st);

struct AttnArgs {
    // self q/k/v/gate: element (b, n, h, d) at ptr[b*bs + n*rs + h*dh + d]
    const float* q;
    const float* k;
    const float* v;
    const float* gate;  // same addressing; out *= sigmoid(gate)
    long bs, rs;
    const float* qw;  // [H][dh] RMSNorm weights (q_norm / k_norm)
    const float* kw;
    float eps;
    const float* rope_cos;  // cos / sin of the angle table [pos][rot_dim] (a0 a0 a1 a1 ..., reference
    const float* rope_sin;  // infer/onnx.py:42-47), precomputed by launch_rope_cossin
    int rot_dim;        // rotated leading dims (64 for DiT, dh for the encoders)
    // cross keys (may be null / 0): [b][h][j][d] contiguous
    const float* k_ref; const float* v_ref; int R;
    const float* k_text; const float* v_text; int P;
    const uint8_t* mask_self;  // [B][N]  key validity (may be null = all valid)
    const uint8_t* mask_ref;   // [B][R]
    const uint8_t* mask_text;  // [B][P]
    int prenormed;      // 1: q and k were already RMS-normalised + rotated in place by launch_qk_prep (0: the kernel does it on the fly)
    float* out; long obs, ors;  // out (b, n, h*dh + d)
    bf16_t* out_hi; bf16_t* out_lo;  // when out_hi != null the result is written as a split bf16 pair instead
    int B, N, H, dh;
};
// fp32 VALU attention (attention.hip): the A/B reference of the DMA + MFMA kernel below (test hook, SMTTS_ATTN_IMG=0)
hipError_t launch_attention(const AttnArgs& a, hipStream_t st);
// In place on the packed projection buffer: q <- RoPE(RMSNorm_head(q) * qw), k <- RoPE(RMSNorm_head(k) * kw)
// (dit.py:95-108); one wave per (row, head, q|k).  Uses the q/k/bs/rs/qw/kw/eps/rope/rot_dim/B/N/H/dh fields.
hipError_t launch_qk_prep(const AttnArgs& a, hipStream_t st);

// ---- attention on producer-written operand images (attention_img.hip) -----------------------------------------------------------
// Every array is a 16-bit GEMM-operand array in format `prec` (PREC_F16 / PREC_BF16: `*_lo` unused; PREC_BF16X3: hi + lo pair).
static inline int pad8(int n) { return (n + 7) & ~7; }
struct AttnImg {
    int prec;
    const bf16_t *q, *q_lo;      // [B][H][N][DHP]   RMSNorm_head * w, RoPE, x 1/sqrt(dh)   (DHP = 64 / 128, pad dims zero)
    const bf16_t *k, *k_lo;      // [B][H][N][DHP]   self keys, normalised + rotated
    const bf16_t *vt, *vt_lo;    // [B][H][DHP][Np]  self values transposed, Np = pad8(N), columns >= N zero
    const bf16_t *g, *g_lo;      // [B*N][H*dh]      sigmoid(gate)
    const bf16_t *kc, *kc_lo;    // [B][H][Cp][DHP]  cross keys of this layer (null: self-attention only), pad rows zero
    const bf16_t *vtc, *vtc_lo;  // [B][H][DHP][Cp]  cross values transposed; Cp = Rp + pad8(P), Rp = pad8(R)
    const uint8_t *mask_self, *mask_ref, *mask_text;   // [B][N], [B][R], [B][P] key validity (null = all valid)
    bf16_t *out_hi, *out_lo;     // o[(b*N + n)*ors + h*dh + d] in the format store_split4 derives from out_lo
    long ors;
    int B, N, H, dh, Np, R, P, Rp, Cp;
};
hipError_t launch_attention_img(const AttnImg& a, hipStream_t st);
// fp32 projection rows [B*N][4*H*dh] = [q | k | v | gate] (bias included) -> the self part of AttnImg (dit.py:95-108; the arithmetic
// the gemm3 EpiQKV epilogue performs on its accumulators, as a stand-alone kernel: test hook + reference for the epilogue)
struct QkvPackArgs {
    const float* qkvg;
    const float *qw, *kw;        // [H][dh]
    float eps, q_scale;
    const float *rope_cos, *rope_sin;   // [pos][rot_dim]
    int rot_dim, prec;
    bf16_t *q, *q_lo, *k, *k_lo, *vt, *vt_lo, *g, *g_lo;
    int B, N, H, dh, dhp, Np;
};
hipError_t launch_qkv_pack(const QkvPackArgs& p, hipStream_t st);
// fp32 cross-KV caches [L][B][H][R|P][dh] (the C ABI's rank-5 tensors) -> Kc [L][B][H][Cp][dhp], Vc^T [L][B][H][dhp][Cp]
struct CrossPackArgs {
    const float *k_ref, *v_ref, *k_text, *v_text;
    bf16_t *kc, *kc_lo, *vtc, *vtc_lo;
    int prec, L, B, H, dh, dhp, R, P, Rp, Cp;
};
hipError_t launch_cross_pack(const CrossPackArgs& p, hipStream_t st);
// out[i] = hi[i] + lo[i]  (a split bf16 pair back to fp32: test hooks)
hipError_t launch_split_to_f32(const bf16_t* hi, const bf16_t* lo, float* out, long n, hipStream_t st);

// out[m][:] = table[ids[m]][:]  (phonemes.py:201)
hipError_t launch_embedding(const int64_t* ids, const float* table, float* out, int M, int C, int vocab, hipStream_t st);

// e[r][i] = sin(1000 t[r] f_i) (i<128) | cos (i>=128), f_i = exp(-i ln(1e4)/127)   (model.py:23-28)
hipError_t launch_time_sinusoid(const float* t, float* e, int rows, hipStream_t st);

// mask[b][j] = j < min(len[b], R)   (style.py:155-162)
hipError_t launch_len_mask(const int64_t* len, uint8_t* mask, int B, int R, hipStream_t st);

// gm[(b*G+g)][pad + t][c] = mask[b][t] ? h[b][t][g*cpg + c] : 0 ; pad frames and c>=cpg are zero.
hipError_t launch_convpos_pack(const float* h, const uint8_t* mask, bf16_t* gm_hi, bf16_t* gm_lo, int B, int T, int G,
                               int cpg, int pad, int gstride, hipStream_t st);
// in place: table[r][c0 + c] = tanh(table[r][c0 + c]) for the two gate column blocks of every DiT block
// LN-fold tables (gemm.hpp LnFoldIn): for every sampler step (modulation-table row) and every AdaLN'd GEMM site of the DiT blocks,
//   tab[row][0][off + n] = sum_k W[n][k] shift[k],   tab[row][1][off + n] = sum_k W[n][k] (1 + scale[k])        (K = 960)
// with W the 16-bit weights the GEMM itself multiplies (fmt 0: one fp16 array, 1: bf16 hi (+ lo when non-null)).
struct FoldSite { const bf16_t* w; const bf16_t* wlo; int N; int fmt; int shift_off, scale_off; long out_off; };
struct FoldSites { FoldSite s[24]; int n; long NF; };
hipError_t launch_fold_vectors(const FoldSites& sites, const float* mod, long mod_ld, int rows, float* tab, hipStream_t st);
hipError_t launch_tanh_gates(float* mod, int rows, long ld, int n_blocks, int per_block, int hidden, hipStream_t st);
// fp32 -> split bf16 pair, rows addressed through RowMaps (element offsets)
hipError_t launch_to_split(const float* x, RowMap xmap, bf16_t* hi, bf16_t* lo, RowMap omap, int M, int C, hipStream_t st);

// sampler element-wise steps (infer/onnx.py:105,125 ; teacher ODE see DESIGN.md)
hipError_t launch_axpby(float* out, const float* x, const float* y, float a, float b, long n, hipStream_t st);
// x0 = a*xt - s*v ; eps = s*xt + a*v ; xt_next = a2*x0 + s2*eps    (x0 written to x0_out)
hipError_t launch_ode_step(float* xt, const float* v, float* x0_out, float a, float s, float a2, float s2, long n,
                           hipStream_t st);
// v = vc + st*(vc - vt) + ss*(vc - vs) over [3][n] stacked velocities (distill.py:101-105)
hipError_t launch_cfg_combine(const float* v3, float* v, float s_text, float s_spk, long n, hipStream_t st);

// Philox4x32-10 + Box-Muller standard normals; element i uses counter (i/4, stream, 0, 0), key (seed lo, hi).
hipError_t launch_randn(float* out, long n, uint64_t seed, uint64_t stream, hipStream_t st);

// synthetic weights (smalltts_amd/weights.py recipe, bit-exact)
hipError_t launch_synth(float* out, long n, uint64_t key, float mean, float half_range, hipStream_t st);
// fp32 -> GEMM weight arrays with optional row permutation/packing: dst row r <- src row perm[r] (perm null = identity).
// (hi, lo): bf16 split pair (or a single array, see common.hpp); h16 (optional): the same rows as fp16 for PREC_F16 sites
hipError_t launch_split_rows(const float* src, long src_ld, bf16_t* hi, bf16_t* lo, long dst_ld, int rows, int cols,
                             const int* perm, hipStream_t st, bf16_t* h16 = nullptr);
// l16 = fp16(src - float(h16)) over [rows][cols] (dst row stride dst_ld): low half of the fp16 weight pair of PREC_F16X2
hipError_t launch_f16_residual(const float* src, long src_ld, const bf16_t* h16, bf16_t* l16, long dst_ld, int rows, int cols, hipStream_t st);
// out2[0] = max_j (sqrt(C) ||W1_j o g|| + |b1_j|), out2[1] = sqrt(C) max |g|   (out2 zeroed by the caller; g / b1 may be null)
hipError_t launch_ffn_range_bound(const float* w1, const float* b1, const float* g, int F, int C, float* out2, hipStream_t st);
hipError_t launch_fill(float* p, float v, long n, hipStream_t st);
hipError_t launch_copy_strided(const float* src, long sld, float* dst, long dld, int rows, int cols, hipStream_t st);
// dst[n][k] (fp32, row-major N x K) = src[base + n1*sn1 + n0*sn0 + k1*sk1 + k0*sk0]  if k0 < k0valid else 0
// with n1 = n / n0dim, n0 = n % n0dim, k1 = k / k0dim, k0 = k % k0dim.  Covers the conv->GEMM weight re-layouts.
struct GatherSpec { long base, sn1, sn0, sk1, sk0; int n0dim, k0dim, k0valid; };
hipError_t launch_gather_pack(const float* src, float* dst, int N, int K, GatherSpec g, hipStream_t st);
// tab[pos][d] = pos * theta^(-(d & ~1)/dim)   pos < npos, d < dim   (rope angle tables; dit.py:138-149, style.py:13-18)
hipError_t launch_rope_table(float* tab, int npos, int dim, hipStream_t st);
// c[i] = cos(ang[i]), s[i] = sin(ang[i])
hipError_t launch_rope_cossin(const float* ang, float* c, float* s, int n, hipStream_t st);

// ---- codec element-wise kernels (channels-last padded images [B][pad + T][C]) -------------------
// x[b][t][c] += gamma[c] * (sum_k w[c][k] * n[b][t - (K-1) + k][c] + bias[c])   (causal depthwise conv)
hipError_t launch_dwconv_resid(float* x, const float* n, const float* w, const float* bias, const float* gamma,
                               int B, int T, int C, int K, int pad, hipStream_t st);
// dwconv_resid + the RMSNorm (weight norm_w) of the updated rows written as a split bf16 pair (the next GEMM's A operand)
// wide-stage codec mixer + the FFN's RMSNorm in one pass (kernels.hip mixer_wide_kernel): xin -> xout (another image), n2 = 16-bit rows [B*T][C]
bool mixer_wide_ok(int C, int K);
hipError_t launch_mixer_wide(const float* xin, float* xout, const float* norm_w, const float* w, const float* bias, const float* gamma,
                             const float* ffn_norm_w, bf16_t* n2hi, bf16_t* n2lo, int B, int T, int C, int K, int pad, float eps,
                             hipStream_t st);
hipError_t launch_dwconv_resid_rms(float* x, const float* n, const float* w, const float* bias, const float* gamma, int B, int T,
                                   int C, int K, int pad, float eps, const float* norm_w, bf16_t* yhi, bf16_t* ylo, RowMap ymap,
                                   hipStream_t st);
// audio[b][t] = bias + sum_{k,c} w[k][c] * x[b][t - (K-1) + k][c]    (head conv, Cout = 1)
hipError_t launch_head_conv(const float* x, const float* w, float bias, float* audio, int B, int T, int C, int K,
                            int pad, hipStream_t st);
// x[b][pad+t][c] = bias[c] + sum_k w[c][k] * audio[b][t-(K-1)+k]   (encoder stem conv, Cin = 1)
hipError_t launch_stem_conv1(const float* audio, const float* w, const float* bias, float* x, int B, int T, int C,
                             int K, int pad, hipStream_t st);
hipError_t launch_zero_pad_frames(float* x, int B, int T, int C, int pad, hipStream_t st);
hipError_t launch_zero_pad_frames3(float* x0, float* x1, float* x2 /* may be null */, int B, int T, int C, int pad, hipStream_t st);

// Fused codec FFN block for C in {32, 64, 128}: x += gamma * (W2 gelu(W1 rmsnorm(x) + b1) + b2), hidden kept in LDS.
// w1 packed [F][CP], w2 packed [CP][F] with CP = max(C, 64) (zero padded).  (codec_ffn.hip)
// C in {32, 64}: weights LDS-resident, wave-autonomous (codec_ffn_wave.hip); w1 [F][ld1], w2 [>= C][F]
hipError_t launch_codec_ffn_wave(float* x, RowMap img, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo, int ld1,
                                 const float* b1, const bf16_t* w2hi, const bf16_t* w2lo, const float* b2, const float* gamma,
                                 int M, int C, int F, float eps, int split, hipStream_t st);
// C in {32, 64}: mixer + FFN of one codec block in ONE pass over the image (codec_ffn_wave.hip, MIX kernels): xout = block(xin)
// all blocks of a C = 32 codec stage in one launch (codec_ffn_wave.hip, stage chain): per block the one-pass block's operands
struct CodecChainBlock {
    const float *mnorm_w, *dw_w, *dw_b, *mgamma, *norm_w;
    const bf16_t* w1; int ld1; const float* b1;
    const bf16_t* w2; const float *b2, *gamma;
};
bool codec_chain_wave_ok(int C, int F, int K, int T, int split, int nb);
hipError_t launch_codec_chain_wave(const float* xin, float* xout, RowMap img, const CodecChainBlock* blocks, int nb, int M, int C, int F, int K,
                                   float eps, int split, hipStream_t st);
bool codec_block_wave_ok(int C, int F, int K, int T, int split);
hipError_t launch_codec_block_wave(const float* xin, float* xout, RowMap img, const float* mnorm_w, const float* dw_w, const float* dw_b,
                                   const float* mgamma, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo, int ld1,
                                   const float* b1, const bf16_t* w2hi, const bf16_t* w2lo, const float* b2, const float* gamma, int M,
                                   int C, int F, int K, float eps, int split, hipStream_t st);
// C in {128, 256}: weights streamed through an LDS ring (codec_ffn_stream.hip); w1 [F][C], w2t = launch_w2_tile_pack(W2 [C][F])
hipError_t launch_codec_ffn_stream(float* x, RowMap img, const float* norm_w, const bf16_t* w1hi, const bf16_t* w1lo,
                                   const float* b1, const bf16_t* w2thi, const bf16_t* w2tlo, const float* b2, const float* gamma,
                                   int M, int C, int F, float eps, int split, hipStream_t st);
hipError_t launch_w2_tile_pack(const bf16_t* in, bf16_t* out, int C, int F, hipStream_t st);



// x[m][n] += mask(m) * gate[(grow0 + (m / rows_per_batch) * grstride) * gld + n] * (sum_s part[s][m][n] + bias[n])
// (gate == null -> 1; fixed summation order s = 0..S-1).  Closes a split-K GEMM (see gemm3_resid_splitk).
// split-K reduce + gated residual, then LayerNorm * (1 + scale) + shift of the updated row -> split bf16 (the next AdaLN)
// Streaming ConvTranspose1d-as-GEMM for the two finest codec stages (W resident in LDS, one wave per 32 rows): codec_upsample.hip
bool codec_upsample_wave_ok(int K, int N);
hipError_t launch_codec_upsample_wave(const float* x, RowMap amap, const bf16_t* whi, const bf16_t* wlo, int ldw, const float* bias,
                                      float* out, RowMap omap, int M, int K, int N, int split, hipStream_t st);
hipError_t launch_splitk_resid_ln(const float* part, int S, float* x, const float* bias, const float* gate, long gld,
                                  int grow0, int grstride, int rows_per_batch, const uint8_t* rowmask, int M, int N, float eps,
                                  const float* shift, const float* scale, bf16_t* yhi, bf16_t* ylo, hipStream_t st,
                                  bool rms = false /* true: y = RMSNorm(x; weight = shift, eps), encoders */);
hipError_t launch_splitk_resid(const float* part, int S, float* x, const float* bias, const float* gate, long gld,
                               int grow0, int grstride, int rows_per_batch, const uint8_t* rowmask, int M, int N,
                               hipStream_t st, const RowMap* xmap = nullptr /* rows of x; default plain [M][N] */,
                               bool overwrite = false /* x = gate * (sum + bias) instead of x += ... */);

// Fused mixer for C <= 256 (out of place: tiles read K-1 halo frames that belong to the neighbouring tile, so the
// update cannot be done in place):  xout[b][t][c] = xin + gamma[c] * (bias[c] + sum_k w[k][c] * n[t-(K-1)+k][c]) with
// n = RMSNorm(xin; g, eps) recomputed in LDS for the tile + halo (x read once, written once).  Both images must have
// zero pad frames.
hipError_t launch_mixer_fused(const float* xin, float* xout, const float* norm_w, const float* w, const float* bias,
                              const float* gamma, int B, int T, int C, int K, int pad, float eps, hipStream_t st);

// device-side audio helpers (kernels.hip): polyphase resampler with a caller-built bank [up][klen], float -> PCM16
hipError_t launch_resample_poly(const float* x, long n_in, const float* bank, int up, int down, int klen, int width, float* y,
                                long n_out, int channels, hipStream_t st);
hipError_t launch_pcm16(const float* x, int16_t* y, long n, hipStream_t st);
// t[i] = float32(np.linspace(1, 0, n))[i] on the device (sampler timesteps)
hipError_t launch_linspace10(float* t, int n, hipStream_t st);
