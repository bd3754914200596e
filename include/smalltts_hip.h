/* smalltts_hip.h — C ABI of libsmalltts_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the four opaque onnxruntime graphs the reference drives
 * (SURVEY.md §8b).  Each entry point cites the reference interface it replaces
 * (paths relative to the reference repository root):
 *
 *   smtts_cond_encode   <- condition_encoder.onnx   src/smalltts/infer/onnx.py:91-96,
 *                                                   src/server/src/pipeline.rs:122-141  (= DiTModel.encode_conditions,
 *                                                   src/smalltts/models/backbone/model.py:88-95)
 *   smtts_denoise_step  <- denoiser.onnx            src/smalltts/infer/onnx.py:107-124,
 *                                                   src/server/src/pipeline.rs:143-166  (= DiTModel.denoise_step, model.py:97-100)
 *   smtts_sample        <- the host-side sampler loop src/smalltts/infer/onnx.py:98-125, pipeline.rs:84-93
 *                          (mode 1: teacher ODE + CFG, built from src/scripts/train/dmd2/distill.py:60-134)
 *   smtts_codec_decode  <- codec/decoder.onnx       src/smalltts/codec/onnx.py:34-53, infer/onnx.py:127-128
 *   smtts_codec_encode  <- codec/encoder.onnx       src/smalltts/codec/onnx.py:56-75
 *
 * Conventions: plain C types only; every tensor argument is a DEVICE pointer on the handle's GPU
 * (e.g. torch tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream);
 * all work is enqueued asynchronously on `stream`.  The caller owns inputs, outputs and workspace
 * (query the size first); the library owns weights only.  Return 0 = ok, non-zero = error with the
 * message available from smtts_last_error().
 *
 * Threading / streams (the same rules as smalltts_amd/csrc/engine.hpp:4-8): a handle is NOT thread-safe — one handle per GPU,
 * driven by ONE host thread (the reference also has one Session per pipeline behind a mutex, src/server/src/main.rs:24,138).
 * That thread MAY keep several operator calls in flight on DIFFERENT streams, provided that
 *   (1) every call in flight has its own workspace and its own output buffers (all per-call scratch lives in the workspace;
 *       weights are read-only after smtts_finalize);
 *   (2) smtts_set_tuning(h, 1) ("throughput") was selected first: kernels that cost the fewest CU-microseconds, no side streams
 *       (the side stream of smtts_cond_encode / smtts_sample belongs to the caller's stream — one per caller stream — so calls in
 *       flight would not share it, but six streams for three batches run 20 % slower than three: profiles/r06j_ab_dual_tp.txt);
 *   (3) per-kernel profiling (smtts_profile_enable) is off: it assumes one call at a time.
 * Calls on one stream are ordered like any other work on that stream; results do not depend on what runs on the other streams
 * (tests/test_api_gpu.py::test_results_repeat_bit_for_bit_next_to_other_streams).
 * bool tensors are 1 byte per element (numpy/torch bool).
 */
#ifndef SMALLTTS_HIP_H
#define SMALLTTS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smtts_engine* smtts_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int smtts_create(int device_id, smtts_handle* out);
int smtts_destroy(smtts_handle h);
const char* smtts_last_error(smtts_handle h); /* h may be NULL: last creation error */
const char* smtts_version(void);
/* bumped on every signature / default change: 5 = round 6 (smtts_test_set_ln_fold; one side stream per caller stream); 4 = round 4 (workspace queries take R and P, new handles default to preset 2,
 * smtts_get_saturations) */
#define SMTTS_ABI_VERSION 5
int smtts_abi_version(void);

/* ---- weights (replaces the ONNX initialisers; names/shapes = DiTModel.state_dict(),
 *      src/scripts/test_checkpoint.py:44-73, plus codec.* names of smalltts_amd/weights.py) ---- */
int smtts_set_tensor(smtts_handle h, const char* name, const float* data, const int64_t* shape, int ndim,
                     int data_on_device);
int smtts_synth_tensor(smtts_handle h, const char* name, const int64_t* shape, int ndim, uint64_t key, float mean,
                       float half_range);
int smtts_get_tensor(smtts_handle h, const char* name, float* host_out, int64_t numel);
/* codec hyper-parameters (decoder order); must precede smtts_finalize when codec tensors are present */
int smtts_set_codec_spec(smtts_handle h, int latent_dim, int n_filters, int kernel, int ffn_mult, float eps,
                         const int* ratios, int n_ratios, const int* depths /* n_ratios + 1 */);
int smtts_finalize(smtts_handle h);
/* GEMM operand precision preset (fp32 accumulation, fp32 residual stream / norms / softmax / sampler state in all of them):
 *   2 = "f16 mixed" — THE DEFAULT of a new handle (smtts_default_precision() == 2; the Python host side sets the same):
 *       ONE fp16 MFMA per product on the DiT-block / encoder / cross-KV / codec-FFN GEMMs (>= 95 % of the flops and
 *       weight bytes), split-bf16 on the conditioning chain, latent in / out projections and codec resampling convs
 *       (measured: latent rel-L2 ~1.5e-4 vs the fp32 oracle, inside the 1e-3 contract)
 *   3 = split-bf16 everywhere: x = hi + lo, three bf16 MFMAs per product (fp32-class results, ~1.4x the time)
 *   1 = single-pass bf16 everywhere (latent rel-L2 ~4e-3: outside the 1e-3 contract, kept for A/B) */
int smtts_set_precision(smtts_handle h, int preset);
int smtts_get_precision(smtts_handle h);   /* the preset in force (1 / 2 / 3) */
int smtts_default_precision(void);         /* host-only: the preset smtts_create starts with */
/* one GEMM site group at a time: site 0 DiT blocks, 1 encoders, 2 cross-KV, 3 conditioning / in / out projections,
 * 4 codec FFNs, 5 codec stem / resampling convs, 6 conv pos-embed, 7 attention operands (q, k, v, gate, probabilities);
 * prec 1 bf16, 2 fp16, 3 split-bf16 (call after smtts_set_precision); site 5 (codec resampling convs) also takes
 * 4 = "fp16 x 2": fp16 activations against fp16 hi + lo weights, two passes instead of split-bf16's three, on the decoder's
 * ConvTranspose stages with 512 <= K <= 1024 (the other stages stay split-bf16) */
int smtts_set_site_precision(smtts_handle h, int site, int prec);
/* fp16 range guard.  fp16 operands saturate at +-65504 instead of overflowing to inf, which is silent; every producer of an
 * fp16 operand therefore counts the values it had to clamp into a per-site device counter (sites as above).  counts[i] = clamps
 * of site i since the last reset; for site 4 additionally the number of fused codec FFN blocks whose hidden / input range could
 * not be certified from the weights at smtts_finalize (those kernels carry no run-time check; smtts_range_report names them).
 * Non-zero = the results of that site are clipped: re-run with smtts_set_site_precision(site, 3) (split-bf16 has fp32 range) —
 * smalltts_amd/api.py does that automatically.  Synchronises the device.  Real checkpoints enter through
 * src/scripts/train/dmd2/distill.py:468-479; the SwiGLU hidden (models/backbone/dit.py:176-186) is the likeliest site. */
int smtts_get_saturations(smtts_handle h, uint32_t* counts, int n_sites, int reset);
const char* smtts_range_report(smtts_handle h);      /* "" when every fused FFN block was certified */
float smtts_range_worst_bound(smtts_handle h);       /* largest certified bound of the last finalize (fp16 max: 65504) */
int smtts_has_part(smtts_handle h, int part); /* 0 dit, 1 codec decoder, 2 codec encoder */

/* ---- condition encoder ---------------------------------------------------------------------- */
size_t smtts_cond_workspace_bytes(smtts_handle h, int B, int R, int P);
/* ref f32 (B,R,64); ref_len i64 (B); phonemes i64 (B,P); ph_mask bool (B,P)
 * -> k_ref,v_ref f32 (12,B,8,R,120); ref_mask bool (B,R); k_text,v_text f32 (12,B,8,P,120).
 * ref_seq_out (B,R,960) / mem_out (B,P,960) are optional debug taps (NULL to skip). */
int smtts_cond_encode(smtts_handle h, void* stream, const float* ref, const int64_t* ref_len, const int64_t* phonemes,
                      const uint8_t* ph_mask, int B, int R, int P, float* k_ref, float* v_ref, uint8_t* ref_mask,
                      float* k_text, float* v_text, void* ws, size_t ws_bytes, float* ref_seq_out, float* mem_out);

/* ---- denoiser --------------------------------------------------------------------------------- */
size_t smtts_denoise_workspace_bytes(smtts_handle h, int B, int N, int R, int P);
/* x_t f32 (B,N,64); mask bool (B,N); t f32 (B); caches as above; rope f32 (1,N,64) angles or NULL
 * -> velocity f32 (B,N,64) */
int smtts_denoise_step(smtts_handle h, void* stream, const float* x_t, const uint8_t* mask, const float* t,
                       const float* k_ref, const float* v_ref, const uint8_t* ref_mask, const float* k_text,
                       const float* v_text, const uint8_t* ph_mask, const float* rope, int B, int N, int R, int P,
                       float* velocity, void* ws, size_t ws_bytes);

/* ---- sampler ---------------------------------------------------------------------------------- */
size_t smtts_sample_workspace_bytes(smtts_handle h, int B, int N, int R, int P, int n_steps, int cfg);
/* mode 0: x=0; for t in linspace(1,0,n): x_t = a x + s eps_i; v = denoise; x = a x_t - s v  (DMD student)
 * mode 1: deterministic ODE from x_1 = s(1) eps (teacher), see DESIGN.md
 * cfg != 0: mask/caches/cond masks carry 3B rows [cond; text dropped; speaker dropped]; x has B rows;
 *           v = vc + s_text (vc - vt) + s_spk (vc - vs).
 * noise: mode 0 (n_steps,B,N,64), mode 1 (B,N,64); NULL -> on-device Philox4x32-10 with `seed`.
 * x_out (B,N,64); steps_out optional (n_steps,B,N,64) x-hat after every step. */
int smtts_sample(smtts_handle h, void* stream, int mode, int n_steps, int cfg, float s_text, float s_spk,
                 const uint8_t* mask, const float* k_ref, const float* v_ref, const uint8_t* ref_mask,
                 const float* k_text, const float* v_text, const uint8_t* ph_mask, int B, int N, int R, int P,
                 const float* noise, uint64_t seed, float* x_out, float* steps_out, void* ws, size_t ws_bytes);

/* ---- codec ------------------------------------------------------------------------------------ */
int smtts_codec_hop(smtts_handle h);
size_t smtts_decode_workspace_bytes(smtts_handle h, int B, int T);
/* latents f32 (B,T,64) -> audio f32 (B,1,hop*T) */
int smtts_codec_decode(smtts_handle h, void* stream, const float* latents, int B, int T, float* audio, void* ws,
                       size_t ws_bytes);
size_t smtts_encode_workspace_bytes(smtts_handle h, int B, int S);
/* audio f32 (B,1,S) -> latents f32 (B, S/hop, 64) */
int smtts_codec_encode(smtts_handle h, void* stream, const float* audio, int B, int S, float* latents, void* ws,
                       size_t ws_bytes);

/* ---- utilities -------------------------------------------------------------------------------- */
/* standard normals, same generator the sampler uses: element i <- Philox4x32-10(counter=(i/4, stream_id), key=seed) */
int smtts_randn(smtts_handle h, void* stream, float* out, int64_t n, uint64_t seed, uint64_t stream_id);
/* (alpha, sigma) of the reference schedule for t (infer/onnx.py:31-39); host-side, float64 math */
void smtts_alpha_sigma(float t, float* alpha, float* sigma);

/* ---- device-side audio front / back end -----------------------------------------------
 * Polyphase windowed-sinc resampling of `channels` rows of n_in samples (reference infer/utils.py:7-16, torchaudio
 * Resample with sinc_interp_kaiser): y[c][f*up + p] = sum_k xpad[c][f*down + k] * bank[p][k], xpad = x with `width`
 * leading zeros.  The caller builds the bank [up][klen] (smalltts_amd/audio.py:_sinc_kernel) and owns every buffer. */
int smtts_resample_poly(smtts_handle h, void* stream, const float* x, int channels, int64_t n_in, const float* bank, int up,
                        int down, int klen, int width, float* y, int64_t n_out);
/* float [-1, 1] -> int16 PCM: clamp, x 32767, round to nearest (reference server audio.rs:22-37; CLIs write PCM_16, tryme.py:29) */
int smtts_pcm16(smtts_handle h, void* stream, const float* x, int64_t n, int16_t* y);

/* cond_encode runs the text encoder on a side stream owned by the engine, one per caller stream (fork / join with events; default
 * on: shortest latency for one batch at a time).  Callers that keep several batches in flight on their own streams should turn it
 * off (throughput tuning does): twice the streams cost more than the overlap buys (7.96 -> 9.59 ms per batch at three in flight). */
int smtts_set_dual_stream(smtts_handle h, int on);
/* Tuning mode: 0 = latency (default: one batch at a time finishes as early as possible: split-K on the small-M projections, deep
 * DMA rings, text encoder on the side stream), 1 = throughput (the caller keeps several independent batches in flight on its own
 * streams: unsplit GEMMs, no side stream, persistent codec kernels on three quarters of the CUs, so that kernels cost the fewest CU-microseconds and leave LDS for the
 * other streams).  Results differ between the modes only by fp32 summation order. */
int smtts_set_tuning(smtts_handle h, int mode);

/* per-kernel HIP-event timing on the launch stream (bench.py roofline): enable, run, then read a JSON array
 * [{"name","launches","ms","flops","bytes"}] of algorithmic work and measured time per kernel class */
int smtts_profile_enable(smtts_handle h, int on);   /* 0 off, 1 per kernel class, 2 + pipeline phase prefix, 3 + GEMM shapes */
int smtts_profile_report(smtts_handle h, char* buf, size_t cap);

/* ---- single-kernel test hooks (used by tests/ to check kernels in isolation) -------------------- */
/* C[M,N] = A[M,K] (f32, lda) * W[N,K]^T (f32 host-layout on device, split internally) + bias ; act as ACT_* */
/* microbenchmark of one GEMM launch configuration (epi: 0 store, 1 store+gelu, 2 swiglu, 3 tanh-gated residual) */
int smtts_bench_gemm(smtts_handle h, int M, int N, int K, int epi, int split, int cfg, int iters, int ver /* kernel generation 1 (fp32 A, register-staged) | 3 (gemm3) */,
                     float* avg_us);
/* hot-path kernel (gemm3: split-bf16 A and W via DMA ring): C[M,N] = act(A W^T + bias); A, W fp32 on device, split internally; K % 64 == 0 */
int smtts_test_gemm3(smtts_handle h, void* stream, const float* A, const float* W, const float* bias, int M, int N, int K,
                     int act, int split, int cfg, float* C);
/* codec blocks: 1 (default) = fused mixer and fused FFN kernels (C <= 256), 0 = separate norm / conv / two-GEMM path */
int smtts_test_set_fused_ffn(smtts_handle h, int on);
/* fused sampler: 1 (default) = the AdaLN between two DiT block GEMMs folded into their epilogues (reference dit.py:19-25,197-212
 * restated as rstd (x (1 + scale) W^T - mu W (1 + scale)) + W shift + b), 0 = split-K reduce + norm (latency) / ln_modulate (throughput) launches.
 * Workspace sizes depend on it: query them after the call. */
int smtts_test_set_ln_fold(smtts_handle h, int on);
int smtts_test_gemm(smtts_handle h, void* stream, const float* A, int lda, const float* W, const float* bias, int M,
                    int N, int K, int act, int split, int cfg, float* C, int ldc);
int smtts_test_swiglu(smtts_handle h, void* stream, const float* A, const float* W1, const float* W3, const float* b1,
                      const float* b3, int M, int F, int K, int split, float* out);
int smtts_test_attention(smtts_handle h, void* stream, const float* qkvg, const float* qw, const float* kw, float eps,
                         const float* rope, int rot_dim, const float* k_ref, const float* v_ref, int R,
                         const float* k_text, const float* v_text, int P, const uint8_t* mask_self,
                         const uint8_t* mask_ref, const uint8_t* mask_text, int B, int N, int H, int dh, float* out);

/* the product's attention path in isolation: stand-alone producers (qkv_pack + cross_pack) write the operand images, then the DMA + MFMA
 * kernel (attention_img.hip) at the site-7 operand precision; same contract as smtts_test_attention (the fp32 VALU reference) */
int smtts_test_attention_mfma(smtts_handle h, void* stream, const float* qkvg, const float* qw, const float* kw, float eps,
                              const float* rope, int rot_dim, const float* k_ref, const float* v_ref, int R,
                              const float* k_text, const float* v_text, int P, const uint8_t* mask_self,
                              const uint8_t* mask_ref, const uint8_t* mask_text, int B, int N, int H, int dh, float* out);
/* engine-wide A/B switch: non-zero (default) = operand images written by the QKVG GEMM epilogue + the DMA / MFMA attention kernel;
 * 0 = fp32 projection + in-place qk_prep + the fp32 VALU reference kernel */
int smtts_test_set_attention_mfma(smtts_handle h, int mode);

#ifdef __cplusplus
}
#endif
#endif
