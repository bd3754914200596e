// smalltts gfx950 engine: owns the weights (fp32 originals + bf16 hi/lo GEMM packs) for one GPU and
// sequences the kernels of the three boundary operators the reference runs through onnxruntime
// (condition_encoder / denoiser / codec, reference infer/onnx.py:91-128) plus the fused sampler.
// Not thread-safe: one Engine per GPU, driven by one host thread (mirrors the reference's one-Session-per-pipeline model,
// src/server/src/main.rs:24).  That thread may keep several operator calls in flight on DIFFERENT streams as long as each
// call has its own workspace and outputs: all per-call scratch (incl. the rope cos / sin of a caller-supplied table) lives
// in the workspace, weights are read-only after finalize().  The dual-stream condition encoder / modulation chain forks onto a side
// stream that belongs to the CALLER's stream (one per caller stream, round 6).  Exception: per-kernel profiling assumes one call at a time.
#pragma once
#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "gemm_ops.hpp"
#include "kernels.hpp"
#include "prof.hpp"

struct RawTensor {
    float* d = nullptr;
    std::vector<long> shape;
    long numel = 0;
};

struct PW {  // packed GEMM weight [N][K]: bf16 hi + lo (PREC_BF16X3 / PREC_BF16) and the same rows as fp16 (PREC_F16)
    bf16_t* hi = nullptr;
    bf16_t* lo = nullptr;
    bf16_t* h16 = nullptr;
    bf16_t* l16 = nullptr;   // fp16(w - float(h16)): only where PREC_F16X2 can run (the codec decoder's ConvTranspose weights)
    int N = 0, K = 0;  // K = row stride = GEMM K (may be zero-padded beyond the source width)
};

struct EncBlockW {
    PW qkvg, wo, ff13, w2;
    const float *qn, *kn, *an, *mn;
};
struct EncoderW {
    int dim, heads, dh, ff, layers;
    float eps;
    std::vector<EncBlockW> blocks;
    const float* final_norm;
    float* rope_cos;  // [MAXPOS][dh] cos / sin of pos * theta^(-2i/dh)
    float* rope_sin;
};
struct DitBlockW {
    std::string name;  // tensor-name prefix of the block ("dit.transformer_blocks.<i>")
    PW qkvg;           // [to_q; to_k_self; to_v_self; gate] unpadded (3840 x 960): only the A/B paths read it (SMTTS_ATTN_IMG=0 / SMTTS_ATTN_EPI=0,
                       // smtts_test_set_attention_mfma) — built on first use (Engine::ensure_qkvg_unpadded), N == 0 until then: 280 MB less resident
    PW out, ff13, ff2;
    PW qkvgp;          // the same rows with every head padded 120 -> 128 (zero rows): [4 x 8 x 128][960], gemm3 EpiQKV's column layout
    float* b_qkvgp;    // bias in that layout (pad and gate entries zero)
    float* b_qkvg;
    const float *b1, *b3, *b2, *qn, *kn;
};
struct CodecBlockW {
    const float *norm_w, *dw_b, *gamma, *ffn_norm_w, *b1, *b2, *ffn_gamma;
    float* dw_w;  // [K][C]
    PW w1, w2;
    PW w2t;       // C = 128 / 256: W2 repacked hidden-tile-major [F/32][C][32] for codec_ffn_stream.hip; N == 0 when unused
    bool f16_ok = true;   // the fused FFN kernels' hidden / input range is certified inside fp16 (Engine::certify_codec_ffn); false:
                          // this block's FFN runs split-bf16 whatever the site precision says (those kernels convert WITHOUT clamping)
};
struct CodecStageW {
    int C = 0, r = 0;  // r: resample ratio entering this stage (0 for stage 0)
    PW resample;       // decoder: ConvTranspose as GEMM [r*C][2*Cprev]; encoder: strided conv [C][2r*Cprev]
    float* resample_bias = nullptr;
    std::vector<CodecBlockW> blocks;
};
struct CodecHalfW {
    bool ready = false;
    std::vector<CodecStageW> stages;
    PW stem;                 // decoder: [C0][K*latent]   (encoder stem is a direct 1-channel conv)
    const float* stem_b = nullptr;
    const float* stem_w_raw = nullptr;  // encoder
    PW head;                 // encoder: [latent][K*Clast]
    float* head_w = nullptr; // decoder: [K][Clast] fp32
    float head_b_host = 0.f;
    const float* head_b = nullptr;
    const float* final_norm_w = nullptr;  // optional RMSNorm in front of the head conv
};

struct CodecSpecC {
    int latent_dim = 64, n_filters = 32, kernel = 7, ffn_mult = 4, n_ratios = 6;
    int ratios[8] = {8, 5, 5, 4, 2, 2, 0, 0};   // decoder order
    int depths[9] = {8, 3, 3, 3, 3, 3, 3, 0, 0};  // decoder order, n_ratios + 1 entries
    float eps = 1e-5f;
    int hop() const { int h = 1; for (int i = 0; i < n_ratios; ++i) h *= ratios[i]; return h; }
};

static constexpr int kDefaultPrecision = PREC_F16;   // "f16 mixed" (DESIGN §2): one default for the C ABI and the Python host side

class Engine {
  public:
    explicit Engine(int device);
    ~Engine();
    int device() const { return device_; }
    const std::string& last_error() const { return err_; }

    // ---- weights ---------------------------------------------------------------------------
    int set_tensor(const char* name, const float* data, const long* shape, int ndim, bool src_on_device);
    int synth_tensor(const char* name, const long* shape, int ndim, uint64_t key, float mean, float half_range);
    int get_tensor(const char* name, float* host_out, long numel);
    int set_codec_spec(const CodecSpecC& s);
    int finalize();  // build GEMM packs for whatever model parts are present
    bool has_dit() const { return dit_ready_; }
    bool has_decoder() const { return dec_.ready; }
    bool has_encoder() const { return enc_.ready; }
    // Per-site operand precision (PREC_* of common.hpp).  Sites: the GEMMs grouped by how much latent / audio error their
    // rounding causes (tests/studies/precision_ladder_cpu.py) and by what they cost.
    enum Site { SITE_DIT_BLOCK = 0,   // QKVG, out-proj, FF1 (SwiGLU), FF2 of the 12 DiT blocks: 95 % of the DiT's flops and bytes
                SITE_ENCODER = 1,     // style / text encoder blocks + their output projections
                SITE_CROSS_KV = 2,    // cross-attention K / V projections of the condition cache
                SITE_COND = 3,        // time / AdaLN modulation chain, latent in-projection, conv pos-embed, velocity head, style in-proj
                SITE_CODEC_FFN = 4,   // codec block FFNs (fused kernels and the wide-stage GEMM pairs)
                SITE_CODEC_CONV = 5,  // codec stem / resampling (ConvTranspose, strided conv) / encoder head GEMMs
                SITE_CONVPOS = 6,     // the grouped conv k = 31 pos-embed of the DiT input embedding (split out of SITE_COND: 1.2 % of a batch's time)
                SITE_ATTN = 7,        // attention operand images (q, k, V^T, sigmoid(gate), P) of the DiT blocks and the encoders
                SITE_COUNT = 8 };
    // preset: 3 = split-bf16 everywhere (fp32-class), 1 = single-pass bf16 everywhere, 2 = "f16 mixed": single-pass fp16 on
    // the block / encoder / cross-KV / codec-FFN GEMMs, split-bf16 on SITE_COND and SITE_CODEC_CONV
    void set_precision(int preset) {
        preset_ = preset == PREC_BF16 ? PREC_BF16 : preset == PREC_F16 ? PREC_F16 : PREC_BF16X3;
        for (int i = 0; i < SITE_COUNT; ++i) prec_[i] = preset_;
        if (preset_ == PREC_F16) prec_[SITE_COND] = prec_[SITE_CODEC_CONV] = prec_[SITE_CONVPOS] = PREC_BF16X3;
    }
    int set_site_precision(int site, int prec) {
        // PREC_F16X2 exists for the codec decoder's ConvTranspose products only (the stages x2_mink_ <= K <= x2_maxk_; the others,
        // the stem and the encoder's strided convs run split-bf16 under it)
        if (site < 0 || site >= SITE_COUNT || prec < 1 || prec > (site == SITE_CODEC_CONV ? PREC_F16X2 : 3))
            return fail("set_site_precision: bad site / precision");
        prec_[site] = prec;
        return 0;
    }
    // Tuning mode.  TUNE_LATENCY (default): one batch at a time should finish as early as possible — split-K on the small-M
    // projections, deep DMA rings, text encoder on the side stream.  TUNE_THROUGHPUT: several independent batches are in flight
    // on the caller's streams (SmallTTS.synthesize_batches, bench.py) — kernels should cost the fewest CU-microseconds and hold
    // the least LDS so that other streams' kernels fit beside them: no split-K, no side stream, persistent codec kernels on three quarters of the CUs (deep rings in both modes since round 3).
    enum { TUNE_LATENCY = 0, TUNE_THROUGHPUT = 1 };
    void set_tuning(int mode);
    int tuning() const { return tuning_; }
    void set_fused_ffn(bool on) { fused_ffn_ = on; }
    void set_attn_img(bool on) { attn_img_ = on; }
    void set_ln_fold(bool on) { ln_fold_ = on; }   // test hook (smtts_test_set_ln_fold): LN-fold on / off, see ln_fold_
    bool ln_fold() const { return ln_fold_; }
    bool attn_img() const { return attn_img_; }
    int site_precision(int site) const { return site >= 0 && site < SITE_COUNT ? prec_[site] : 0; }
    void set_dual_stream(bool on) { dual_stream_ = on; }
    int precision() const { return preset_; }

    // ---- operators (device pointers, async on `st`) ------------------------------------------
    size_t cond_ws_bytes(int B, int R, int P) const;
    int cond_encode(hipStream_t st, const float* ref, const int64_t* ref_len, const int64_t* ids,
                    const uint8_t* ph_mask, int B, int R, int P, float* k_ref, float* v_ref, uint8_t* ref_mask,
                    float* k_text, float* v_text, void* ws, size_t ws_bytes, float* ref_seq_out, float* mem_out);

    size_t denoise_ws_bytes(int B, int N, int R, int P, int rows) const;
    int denoise_step(hipStream_t st, const float* x_t, const uint8_t* mask, const float* t, const float* k_ref,
                     const float* v_ref, const uint8_t* ref_mask, const float* k_text, const float* v_text,
                     const uint8_t* ph_mask, const float* rope, int B, int N, int R, int P, float* velocity,
                     void* ws, size_t ws_bytes);

    // mode 0: DMD re-noising loop (infer/onnx.py:98-125); mode 1: teacher ODE (build-defined).
    // cfg != 0: caches/masks hold 3B rows [cond; no-text; no-speaker], x has B rows.
    // noise: mode 0 -> (n_steps, B, N, 64); mode 1 -> (B, N, 64); null -> Philox(seed).
    size_t sample_ws_bytes(int B, int N, int R, int P, int n_steps, int cfg) const;
    int sample(hipStream_t st, int mode, int n_steps, int cfg, float s_text, float s_spk, const uint8_t* mask,
               const float* k_ref, const float* v_ref, const uint8_t* ref_mask, const float* k_text,
               const float* v_text, const uint8_t* ph_mask, int B, int N, int R, int P, const float* noise,
               uint64_t seed, float* x_out, float* steps_out, void* ws, size_t ws_bytes);

    size_t decode_ws_bytes(int B, int T) const;
    int codec_decode(hipStream_t st, const float* latents, int B, int T, float* audio, void* ws, size_t ws_bytes);
    size_t encode_ws_bytes(int B, int S) const;
    int codec_encode(hipStream_t st, const float* audio, int B, int S, float* latents, void* ws, size_t ws_bytes);

    const CodecSpecC& codec_spec() const { return cspec_; }
    // per-kernel HIP-event timing (bench.py); report = JSON array, valid after the stream is synchronised
    void profile_enable(int mode);  // 0 off, 1 per-kernel, 2 per-kernel prefixed with the pipeline phase
    std::string profile_report();
    // single-kernel hooks for tests (W* are fp32 [N][K] on the device; split here, freed after the call)
    int test_gemm(hipStream_t st, const float* A, int lda, const float* W, const float* bias, int M, int N, int K,
                  int act, int split, int cfg, float* C, int ldc);
    int test_swiglu(hipStream_t st, const float* A, const float* W1, const float* W3, const float* b1, const float* b3,
                    int M, int F, int K, int split, float* out);

    // microbenchmark: time `iters` launches of one GEMM configuration with HIP events (tools/gemm_bench.py)
    int bench_gemm(int M, int N, int K, int epi, int split, int cfg, int iters, int ver, float* avg_us);
    int test_gemm3(hipStream_t st, const float* A, const float* W, const float* bias, int M, int N, int K, int act,
                   int split, int cfg, float* C);

    // fp16 range guard: per-site count of values that fp16 producers clamped to +-65504 since the last reset (device counters,
    // common.hpp sat_note) + the static part: codec FFN blocks whose fused kernels' hidden / input bound could not be certified
    // below the fp16 range at finalize() (those kernels do not count at run time: they are VALU-bound, see finalize_codec).
    // Synchronises the device.
    int get_saturations(unsigned* out, int n, bool reset);
    const std::string& range_report() const { return range_report_; }
    float range_worst() const { return range_worst_; }

    int fail(const std::string& m) { err_ = m; return 1; }
    int fail_hip(hipError_t e, const char* what);

  private:
    void* dalloc(size_t bytes);
    const RawTensor* raw(const std::string& n) const;
    const float* rawp(const std::string& n) const;
    PW pack_rows(const std::vector<std::string>& names, const std::vector<int>* perm = nullptr, int k_pad = 0);
    PW pack_from_f32(const float* src, int N, int K, int k_pad = 0, int n_pad = 0);
    float* concat_vec(const std::vector<std::string>& names, const std::vector<int>& zero_len = {});
    int finalize_dit();
    int finalize_codec(bool decoder);
    int build_encoder(EncoderW& e, const std::string& prefix, int dim, int heads, int ff, int layers, float eps);
    int run_encoder(hipStream_t st, const EncoderW& e, void* enc_ws, int B, int S, const uint8_t* key_mask);
    int make_rope(int dim, float** cos_out, float** sin_out);
    int modulation(hipStream_t st, const float* t_dev, int rows, float* sinb, float* t1, float* temb, float* e1,
                   float* semb, float* mod, float* ftab = nullptr);
    struct DenoiseWs;
    // cross-KV cache of all layers in the attention kernel's operand format (attention_img.hip), built once per sampler call
    struct CrossImg { bf16_t *kc = nullptr, *kc_lo = nullptr, *vtc = nullptr, *vtc_lo = nullptr; int Rp = 0, Cp = 0; };
    size_t cross_img_bytes(int B, int R, int P) const;
    int pack_cross(hipStream_t st, const float* k_ref, const float* v_ref, const float* k_text, const float* v_text, int B, int R,
                   int P, char* ws, CrossImg& ci);
    int denoise_core(hipStream_t st, const float* x_t, const uint8_t* mask, const float* mod, int mod_row0,
                     int mod_rstride, const float* k_ref, const float* v_ref, const uint8_t* ref_mask,
                     const float* k_text, const float* v_text, const uint8_t* ph_mask, const float* rope, int B,
                     int N, int R, int P, float* velocity, char* ws, const CrossImg& ci, const float* ftab = nullptr);
    size_t denoise_core_bytes(int B, int N) const;
    // runs one block; the result lives in *x on return (the fused mixer ping-pongs *x <-> *xalt)
    int ensure_qkvg_unpadded();   // packs DitBlockW::qkvg of every block on first use of an A/B attention path
    bool qkvg_unpadded_ready_ = false;   // ... set only after EVERY block packed (a failure half-way unlinks what was built)
    int codec_stage_chain(hipStream_t st, const CodecStageW& sg, float** x, float** xalt, int B, int T, int C);
    int codec_block(hipStream_t st, const CodecBlockW& w, float** x, float** xalt, float* nbuf, bf16_t* n2hi, bf16_t* n2lo,
                    bf16_t* hhi, bf16_t* hlo, int B, int T, int C, size_t n2_elems /* capacity of n2hi (bf16 elements) */);
    int check_shape(const std::string& name, std::initializer_list<long> want);
    void invalidate() { finalized_ = false; dit_ready_ = false; dec_.ready = false; enc_.ready = false; }
    void free_packs();

    // the tag a producer of attention operand images is handed in place of the (unused) lo array when the images are fp16
    bf16_t* img_lo(int pa, bf16_t* lo) const { return pa == PREC_F16 ? sm_lo_for(PREC_F16, nullptr, satp(SITE_ATTN)) : lo; }
    int certify_codec_ffn(const std::string& name, CodecBlockW& b, int C, int F);
    unsigned* sat_ = nullptr;                 // [SITE_COUNT] device counters (null if the allocation failed: nothing is counted)
    unsigned* satp(int site) const { return sat_ ? sat_ + site : nullptr; }
    unsigned sat_static_[SITE_COUNT] = {};    // uncertified fused-kernel blocks (set by finalize)
    float* cert_scratch_ = nullptr;           // 2 floats: max hidden bound, max input bound
    std::string range_report_;
    float range_worst_ = 0.f;                 // largest certified bound of the last finalize (diagnostic)

    int device_;
    std::string err_;
    std::map<std::string, RawTensor> raw_;
    std::vector<void*> allocs_;       // raw fp32 tensors (live as long as the engine)
    std::vector<void*> pack_allocs_;  // everything finalize() builds: freed and rebuilt by the next finalize()
    bool packing_ = false;
    int tuning_ = TUNE_LATENCY;
    int gemm_deep_ = 1;   // gemm3 ring depth of this engine's launches (1 deep — both tunings since round 3; 0 shallow: SMTTS_GEMM_DEEP / SMTTS_GEMM_DEEP_TP); installed per operator call (DeepScope)
    int persist_cus_ = 0;        // grid cap of the persistent codec kernels for this engine's calls (0 = one workgroup per CU): 0 under latency tuning,
    int persist_cus_tp_ = 192;   // this under throughput tuning (SMTTS_PERSIST_CUS; profiles/r03ac_*, r03ad_*)
    bool dual_stream_latency_ = true;  // the dual-stream setting that TUNE_LATENCY restores
    int preset_ = kDefaultPrecision;   // set_precision(kDefaultPrecision) in the constructor fills prec_
    int prec_[SITE_COUNT] = {3, 3, 3, 3, 3, 3, 3, 3};
    bool fused_ffn_ = true;  // test hook: smtts_test_set_fused_ffn
    int chain_min_run_ = 16;   // ... and decodes of at least this many tiles per wave of the chain's grid (codec_stage_chain)
    int chain_min_blocks_ = 2; // ... for stages of at least this many blocks (a single block gains nothing from the chain's contiguous walk; SMTTS_CHAIN_MIN=1: debugging)
    bool stage_chain_ = true; // codec stages with C = 32: all blocks of the stage in ONE launch (SMTTS_STAGE_CHAIN=0: one launch per block)
    bool block_wave_ = true;  // codec stages with C = 32 / 64: mixer + FFN in one kernel (SMTTS_BLOCK_WAVE=0: mixer_fused + codec_ffn_wave)
    bool mixer_wide_ = true;   // codec blocks of the wide stages: mixer + FFN norm in one pass (SMTTS_MIXER_WIDE=0: rmsnorm + dwconv_resid_rms)
    int x2_mink_ = 512, x2_maxk_ = 1024;   // PREC_F16X2 on SITE_CODEC_CONV: the ConvTranspose stages with K in this range (SMTTS_X2_MINK / _MAXK)
    int up_g3_mink_ = 2048;  // codec ConvTranspose-as-GEMM: gemm3 on a converted copy of the image from this K up (SMTTS_UP_G3_MINK; below: fp32-A kernel)
    int ksplit_enc_ = 4;  // split-K of the encoders' residual projections (1 = fused-epilogue GEMM + separate RMSNorm)
    int ksplit_out_ = 2, ksplit_ff2_ = 2;  // (<= kSplitK) split-K factors of the two N = 960 DiT projections in latency tuning (1 = fused epilogue).  Round 5
                                           // (profiles/r05h_splitk_table.txt, three interleaved repeats on one box): 2 slices 11.75-11.79 ms per batch one at a
                                           // time, 3 slices (rounds 1-4: 450 workgroups = one round at 2 per CU) 11.86-11.87, unsplit + ln_modulate 11.90-11.98 —
                                           // a launch of these 600-row products is ~8 us of fixed cost whatever its k-loop (5 k-tiles at 3 slices), so the third
                                           // slice only adds a partial slab (a third of the fp32 slab traffic of the reduce kernel)
    bool dual_stream_ = true;  // cond_encode: text encoder on a side stream (SMTTS_SINGLE_STREAM=1 turns it off)
    struct AuxSet { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
    std::map<hipStream_t, AuxSet> aux_sets_;   // side stream + fork / join events per caller stream (ensure_aux)
    hipStream_t aux_ = nullptr;                // ... of the call being enqueued
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    int ensure_aux(hipStream_t st);
    bool ws_ready_ = false, ws_keep_ = false;   // sample(): the denoiser workspace's never-written regions were zeroed by an earlier step of this call
    bool join_pending_ = false;   // sample(): the side stream's modulation table must be joined before the first AdaLN
    int num_cus_ = 256;
    bool convpos_by_group_ = true;  // grouped conv pos-embed as one product per group over the batch's rows (false: per (utterance, group))
    bool attn_img_ = true;   // attention on producer-written operand images (attention_img.hip: DMA + MFMA only); false (SMTTS_ATTN_IMG=0, test hook) = fp32 projection + qk_prep + the fp32 VALU reference kernel (attention.hip)
    // LN-fold (round 6; gemm.hpp LnFoldIn): inside the fused sampler (one modulation row per step for the whole batch) the AdaLN in front
    // of the QKVG / FF1 products is folded into the producing out-proj / FF2 epilogue (operand image + row partials) and the consuming
    // epilogue (mean / rstd correction): no split-K partials + reduce launches in latency tuning — 5 launches per DiT block (and per
    // encoder block: RMSNorm variant, no tables) instead of 7 (SMTTS_LN_FOLD=0 / smtts_test_set_ln_fold restore the norm launches)
    bool ln_fold_ = true;
    // ... under THROUGHPUT tuning too.  Off: with batches in flight the DiT fold's per-call table kernel (fold_vectors: every block weight
    // read once more, 205 MB = 80 us of the whole chip) costs more than the ln_modulate launches it replaces, and the encoders' unsplit
    // 16-workgroup products no less than their split-K 4 + reduce pairs — measured (profiles/r06i_ab_fold.txt, three interleaved repeats):
    // in flight 8.19 (off) / 8.22 (encoders only) / 8.26 (all) ms per batch, one at a time 11.77 -> 11.62 (SMTTS_LN_FOLD_TP=1: A/B)
    bool ln_fold_tp_ = false;
    bool ln_fold_now() const { return ln_fold_ && (tuning_ == TUNE_LATENCY || ln_fold_tp_); }
    bool attn_epi_ = true;   // ... written by the QKVG GEMM's own epilogue (gemm3 EpiQKV); false (SMTTS_ATTN_EPI=0): fp32 projection + qkv_pack kernel
    Profiler prof_;
    bool prof_on_ = false;
    bool finalized_ = false;

    // DiT packs
    bool dit_ready_ = false;
    PW time0_, time2_, emb0_, emb2_, modall_, inproj_, conv1_, conv2_, velocity_, phproj_, kvref_, kvtext_, style_in_,
        style_out_;
    float *modall_b_ = nullptr, *kvref_b_ = nullptr, *kvtext_b_ = nullptr, *knc_ = nullptr;
    std::vector<DitBlockW> blocks_;
    EncoderW style_, text_;
    float style_scale_ = 1.f;
    float *rope_dit_cos_ = nullptr, *rope_dit_sin_ = nullptr;  // [MAXPOS][64]

    CodecSpecC cspec_;
    CodecHalfW dec_, enc_;
};

void alpha_sigma_host(float t, float& a, float& s);

static constexpr int kMaxPos = 4096;   // reference rope tables (dit.py:139, style.py:140)
static constexpr int kHidden = 960, kHeads = 8, kDh = 120, kBlocks = 12, kFF = 2400, kLatent = 64;
static constexpr int kSplitK = 4;   // K slices of the DiT out-proj / FF2 GEMMs
static constexpr int kFFp = 2432;  // FF hidden row stride: 2400 padded to a multiple of 64 (zero tail) for the DMA GEMM
static constexpr int kModPerBlock = 6 * kHidden;
static constexpr long kModLd = (long)kBlocks * kModPerBlock + 2 * kHidden;  // 71040
static constexpr int kConvK = 31, kConvG = 16, kConvCpg = 60, kConvPad = 15, kConvGs = 64;
static constexpr int kCodecPad = 8;
static constexpr long kFoldPerBlock = 4L * kHeads * 128 + 2L * kFF;   // LN-fold table columns of one block: padded QKVG rows | interleaved [w1 | w3] rows
static constexpr long kFoldNF = kBlocks * kFoldPerBlock;              // 106752
static constexpr long kFoldMaxRows = 1024;                            // LN-fold of the DiT blocks up to this many rows (above: unsplit GEMM + ln_modulate, as before)
static constexpr int kLnGroups = kHidden / 32;                        // row partials per residual row
