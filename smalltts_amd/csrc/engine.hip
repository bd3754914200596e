// Engine: weight registry, GEMM packing, and the kernel sequences of the hot-path operators.
#include "engine.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

#define HIPC(x)                                          \
    do {                                                 \
        hipError_t e__ = (x);                            \
        if (e__ != hipSuccess) return fail_hip(e__, #x); \
    } while (0)

namespace {
struct Bump {  // bump allocator over a caller-owned workspace (also used, with p == null, to size it)
    char* p;
    size_t off = 0;
    explicit Bump(void* base) : p(static_cast<char*>(base)) {}
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* r = reinterpret_cast<T*>(p + off);
        off += n * sizeof(T);
        return r;
    }
};
inline std::string sidx(const std::string& a, int i, const std::string& b) { return a + std::to_string(i) + b; }
}  // namespace

int g_gemm3_t160 = 1;   // gemm3 160x128 tiles for M = 600 x wide N (SMTTS_GEMM_T160=0: off)
thread_local int g_gemm3_deep = 1;   // gemm3 ring depth for single-array operand formats: 1 = deep (latency tuning), 0 = shallow (throughput tuning);
                                      // thread-local, installed from the calling Engine for the duration of each operator call (DeepScope)
namespace { struct DeepScope { int prev; explicit DeepScope(int v) : prev(g_gemm3_deep) { g_gemm3_deep = v; } ~DeepScope() { g_gemm3_deep = prev; } }; }
int g_gemm3_w4_minm = 0;  // gemm3: 4-wave 128x128 tiles (wave 64x64) for single-array products with M >= this and N >= 2048 (SMTTS_GEMM_W4_MINM; 0 = off)
int g_gemm3_stage16 = 1;  // gemm3: 16-bit outputs through the LDS-staged epilogue (SMTTS_GEMM_STAGE16=0: scalar stores)
// Persistent codec kernels (streamed / one-pass FFN, upsample): workgroups launched at most (0 = one per CU).  A persistent grid
// that covers every CU keeps all other streams' kernels out until it ends; with several batches in flight (throughput tuning) it
// leaves a quarter of the CUs to them (Engine::persist_cus_, installed per codec call like the ring depth).
thread_local int g_persist_cus = 0;
int g_persist_mask = 7;   // 1 streamed FFN, 2 one-pass / wave FFN, 4 upsample (SMTTS_PERSIST_MASK)
namespace { struct PersistScope { int prev; explicit PersistScope(int v) : prev(g_persist_cus) { g_persist_cus = v; } ~PersistScope() { g_persist_cus = prev; } }; }
int g_gemm_xcd = 1;     // fp32-A GEMM: XCD-aware tile order for unbatched launches (SMTTS_GEMM_XCD=0: plain blockIdx mapping)
int g_gemm3_group = 4;  // gemm3 tile order when neither operand fits an XCD's L2: bands of this many row tiles (SMTTS_GEMM_GROUP=1: plain N fastest)
int g_gemm3_nfast = 1;  // gemm3 tile order: N fastest when M > N (SMTTS_GEMM_NFAST=0 restores M fastest everywhere)
thread_local Profiler* g_prof = nullptr;
thread_local const char* g_prof_tag = nullptr;
thread_local int g_prof_shapes = 0;

static int g_small_m_splitk = 1;   // SMTTS_SMALLM_SPLITK=0: A/B switch for the K-sliced small-M products of the codec
Engine::Engine(int device) : device_(device) {
    set_precision(kDefaultPrecision);
    if (const char* nf = lab_env("SMTTS_GEMM_NFAST")) g_gemm3_nfast = atoi(nf);
    if (const char* gg = getenv("SMTTS_GEMM_GROUP")) g_gemm3_group = atoi(gg);
    if (const char* gx = getenv("SMTTS_GEMM_XCD")) g_gemm_xcd = atoi(gx);
    if (const char* pc = getenv("SMTTS_PERSIST_CUS")) persist_cus_tp_ = atoi(pc);
    if (const char* pm = lab_env("SMTTS_PERSIST_MASK")) g_persist_mask = atoi(pm);
    if (const char* dp = getenv("SMTTS_GEMM_DEEP")) gemm_deep_ = atoi(dp);
    if (const char* s16 = lab_env("SMTTS_GEMM_STAGE16")) g_gemm3_stage16 = atoi(s16);
    if (const char* w4 = lab_env("SMTTS_GEMM_W4_MINM")) g_gemm3_w4_minm = atoi(w4);
    if (const char* t1 = lab_env("SMTTS_GEMM_T160")) g_gemm3_t160 = atoi(t1);
    const char* s = getenv("SMTTS_SINGLE_STREAM");
    if (s && *s == '1') dual_stream_ = false;
    if ((s = lab_env("SMTTS_BLOCK_WAVE"))) block_wave_ = atoi(s) != 0;
    if ((s = getenv("SMTTS_STAGE_CHAIN"))) { stage_chain_ = atoi(s) != 0; if (atoi(s) >= 2) chain_min_run_ = 0; }   // 2: the chain whatever the run length (tests)
    if ((s = lab_env("SMTTS_CHAIN_MIN"))) chain_min_blocks_ = atoi(s);
    if ((s = lab_env("SMTTS_CHAIN_MIN_RUN"))) chain_min_run_ = atoi(s);
    if ((s = lab_env("SMTTS_SMALLM_SPLITK"))) g_small_m_splitk = atoi(s);
    if ((s = lab_env("SMTTS_CONVPOS_BY_GROUP"))) convpos_by_group_ = atoi(s) != 0;
    if ((s = getenv("SMTTS_ATTN_EPI"))) attn_epi_ = atoi(s) != 0;
    if ((s = lab_env("SMTTS_ATTN_IMG"))) attn_img_ = atoi(s) != 0;
    if ((s = getenv("SMTTS_LN_FOLD"))) ln_fold_ = atoi(s) != 0;
    if ((s = lab_env("SMTTS_LN_FOLD_TP"))) ln_fold_tp_ = atoi(s) != 0;
    if ((s = lab_env("SMTTS_UP_G3_MINK")) && atoi(s) >= 64) up_g3_mink_ = atoi(s);
    if ((s = getenv("SMTTS_MIXER_WIDE"))) mixer_wide_ = atoi(s) != 0;
    if ((s = lab_env("SMTTS_X2_MINK")) && atoi(s) >= 64) x2_mink_ = atoi(s);
    if ((s = lab_env("SMTTS_X2_MAXK")) && atoi(s) >= 64) x2_maxk_ = atoi(s);
    if ((s = lab_env("SMTTS_KSPLIT_OUT")) && atoi(s) >= 1 && atoi(s) <= kSplitK) ksplit_out_ = atoi(s);
    if ((s = lab_env("SMTTS_KSPLIT_ENC")) && atoi(s) >= 1 && atoi(s) <= kSplitK) ksplit_enc_ = atoi(s);
    if ((s = lab_env("SMTTS_KSPLIT_FF2")) && atoi(s) >= 1 && atoi(s) <= kSplitK) ksplit_ff2_ = atoi(s);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) num_cus_ = cus;
    // three quarters of the CUs, in whole rounds of the 32 shader engines the dispatcher deals workgroups to (measured in flight,
    // ms per batch at 256 / 224 / 208 / 192 / 176 / 160 / 128 workgroups: 8.62 / 8.46 / 8.55 / 8.41 / 8.49 / 8.44 / 8.54)
    if (!getenv("SMTTS_PERSIST_CUS")) persist_cus_tp_ = num_cus_ >= 64 ? num_cus_ * 3 / 4 / 32 * 32 : 0;
    // fp16 range guard: one saturation counter per precision site (+ two floats of scratch for finalize's certificates)
    if (hipSetDevice(device) == hipSuccess) {
        sat_ = static_cast<unsigned*>(dalloc((SITE_COUNT + 2) * sizeof(unsigned)));
        if (sat_ && hipMemset(sat_, 0, (SITE_COUNT + 2) * sizeof(unsigned)) != hipSuccess) sat_ = nullptr;
        if (sat_) cert_scratch_ = reinterpret_cast<float*>(sat_ + SITE_COUNT);
    }
}

int Engine::get_saturations(unsigned* out, int n, bool reset) {
    HIPC(hipSetDevice(device_));
    HIPC(hipDeviceSynchronize());
    unsigned dev[SITE_COUNT] = {};
    if (sat_) HIPC(hipMemcpy(dev, sat_, sizeof dev, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) out[i] = i < SITE_COUNT ? dev[i] + sat_static_[i] : 0u;
    if (reset && sat_) HIPC(hipMemset(sat_, 0, sizeof dev));   // (the static part is a property of the weights: it stays)
    return 0;
}

void Engine::set_tuning(int mode) {
    if (mode == TUNE_THROUGHPUT) {
        if (tuning_ != TUNE_THROUGHPUT) dual_stream_latency_ = dual_stream_;
        tuning_ = TUNE_THROUGHPUT;
        // (A/B only: side streams with batches in flight.  Round 6 gave every caller stream its OWN side stream — three batches in flight
        // then drive six streams, and the batch goes from 7.96 to 9.59 ms (profiles/r06j_ab_dual_tp.txt): more streams than hardware
        // queues serialise worse than one text encoder behind its style encoder.  Stays off.)
        dual_stream_ = lab_env("SMTTS_DUAL_TP") && atoi(lab_env("SMTTS_DUAL_TP")) != 0;
        // Ring depth with batches in flight: round 2 measured shallow rings ahead (10.14 vs 10.38 ms: a workgroup holding 64-128 KiB
        // of LDS while it waits kept the other streams' kernels off its CU).  Re-measured at the end of round 3 — fp16 operand images,
        // shorter epilogues, persistent codec grids capped — deep rings win: 8.53 -> 8.42 ms (profiles/r03ag_*).
        gemm_deep_ = getenv("SMTTS_GEMM_DEEP") ? atoi(getenv("SMTTS_GEMM_DEEP")) : 1;
        persist_cus_ = persist_cus_tp_;
    } else {
        if (tuning_ == TUNE_THROUGHPUT) dual_stream_ = dual_stream_latency_;
        tuning_ = TUNE_LATENCY;
        persist_cus_ = 0;
        gemm_deep_ = getenv("SMTTS_GEMM_DEEP") ? atoi(getenv("SMTTS_GEMM_DEEP")) : 1;
    }
}

void Engine::profile_enable(int mode) {
    const bool on = mode != 0;
    prof_on_ = on;
    prof_.reset();
    g_prof = on ? &prof_ : nullptr;
    g_prof_tag = mode >= 2 ? "" : nullptr;  // 2 = tagged: kernel names carry the pipeline phase
    g_prof_shapes = mode == 3;               // 3 = tagged + the GEMM classes carry their product's shape
}

std::string Engine::profile_report() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    auto agg = prof_.collect();
    std::string out = "[";
    bool first = true;
    char buf[512];
    for (auto& kv : agg) {
        snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"bytes8d\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.flops, kv.second.bytes, kv.second.bytes8d);
        out += buf;
        first = false;
    }
    prof_.reset();
    return out + "]";
}

Engine::~Engine() {
    if (g_prof == &prof_) g_prof = nullptr;
    (void)hipSetDevice(device_);
    for (auto& kv : aux_sets_) {
        if (kv.second.stream) (void)hipStreamDestroy(kv.second.stream);
        if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
        if (kv.second.join) (void)hipEventDestroy(kv.second.join);
    }
    for (void* p : allocs_) (void)hipFree(p);
    for (void* p : pack_allocs_) (void)hipFree(p);
}

int Engine::fail_hip(hipError_t e, const char* what) {
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}

void* Engine::dalloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    (packing_ ? pack_allocs_ : allocs_).push_back(p);
    return p;
}

void Engine::free_packs() {
    for (void* p : pack_allocs_) (void)hipFree(p);
    pack_allocs_.clear();
}

int Engine::check_shape(const std::string& name, std::initializer_list<long> want) {
    const RawTensor* t = raw(name);
    if (!t) return fail("missing tensor " + name);
    if (t->shape.size() == want.size() && std::equal(want.begin(), want.end(), t->shape.begin())) return 0;
    auto str = [](auto b, auto e) { std::string o = "("; for (auto i = b; i != e; ++i) o += (i == b ? "" : ", ") + std::to_string(*i); return o + ")"; };
    return fail("tensor " + name + " has shape " + str(t->shape.begin(), t->shape.end()) + ", this build expects " +
                str(want.begin(), want.end()));
}

const RawTensor* Engine::raw(const std::string& n) const {
    auto it = raw_.find(n);
    return it == raw_.end() ? nullptr : &it->second;
}
const float* Engine::rawp(const std::string& n) const {
    const RawTensor* t = raw(n);
    return t ? t->d : nullptr;
}

// ---------------------------------------------------------------------------------------------
// weights in
// ---------------------------------------------------------------------------------------------
int Engine::set_tensor(const char* name, const float* data, const long* shape, int ndim, bool src_on_device) {
    HIPC(hipSetDevice(device_));
    RawTensor t;
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
    auto it = raw_.find(name);
    if (it != raw_.end() && it->second.numel == t.numel) {
        t.d = it->second.d;
    } else {
        t.d = static_cast<float*>(dalloc(t.numel * sizeof(float)));
        if (!t.d) return fail("out of device memory for tensor " + std::string(name));
    }
    HIPC(hipMemcpy(t.d, data, t.numel * sizeof(float), src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    raw_[name] = t;
    invalidate();
    return 0;
}

int Engine::synth_tensor(const char* name, const long* shape, int ndim, uint64_t key, float mean, float half_range) {
    HIPC(hipSetDevice(device_));
    RawTensor t;
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
    t.d = static_cast<float*>(dalloc(t.numel * sizeof(float)));
    if (!t.d) return fail("out of device memory for tensor " + std::string(name));
    HIPC(launch_synth(t.d, t.numel, key, mean, half_range, 0));
    raw_[name] = t;
    invalidate();
    return 0;
}

int Engine::get_tensor(const char* name, float* host_out, long numel) {
    const RawTensor* t = raw(name);
    if (!t) return fail("unknown tensor " + std::string(name));
    if (numel != t->numel) return fail("size mismatch reading tensor " + std::string(name));
    HIPC(hipSetDevice(device_));
    HIPC(hipMemcpy(host_out, t->d, numel * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int Engine::set_codec_spec(const CodecSpecC& s) {
    if (s.n_ratios < 1 || s.n_ratios > 7) return fail("codec spec: 1..7 ratios supported");
    cspec_ = s;
    invalidate();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// packing helpers
// ---------------------------------------------------------------------------------------------
PW Engine::pack_from_f32(const float* src, int N, int K, int k_pad, int n_pad) {
    PW w;
    const int Kp = k_pad > K ? k_pad : K;
    const int Np = n_pad > N ? n_pad : N;  // extra all-zero rows (allocation only; w.N stays the logical row count)
    w.N = N;
    w.K = Kp;
    w.hi = static_cast<bf16_t*>(dalloc((size_t)Np * Kp * 2));
    w.lo = static_cast<bf16_t*>(dalloc((size_t)Np * Kp * 2));
    w.h16 = static_cast<bf16_t*>(dalloc((size_t)Np * Kp * 2));
    if (!w.hi || !w.lo || !w.h16) { w.N = 0; return w; }
    if (Kp != K || Np != N) {
        (void)hipMemsetAsync(w.hi, 0, (size_t)Np * Kp * 2, 0);
        (void)hipMemsetAsync(w.lo, 0, (size_t)Np * Kp * 2, 0);
        (void)hipMemsetAsync(w.h16, 0, (size_t)Np * Kp * 2, 0);
    }
    if (launch_split_rows(src, K, w.hi, w.lo, Kp, N, K, nullptr, 0, w.h16) != hipSuccess) w.N = 0;
    return w;
}

PW Engine::pack_rows(const std::vector<std::string>& names, const std::vector<int>* perm, int k_pad) {
    PW w;
    int K = -1, Ntot = 0;
    for (auto& n : names) {
        const RawTensor* t = raw(n);
        if (!t || t->shape.size() != 2) { err_ = "pack: missing/non-2D tensor " + n; return w; }
        if (K < 0) K = (int)t->shape[1];
        if (K != t->shape[1]) { err_ = "pack: K mismatch at " + n; return w; }
        Ntot += (int)t->shape[0];
    }
    if (names.size() == 1 && !perm) return pack_from_f32(raw(names[0])->d, Ntot, K, k_pad);
    float* tmp = nullptr;
    if (hipMalloc(&tmp, (size_t)Ntot * K * 4) != hipSuccess) { err_ = "pack: temp alloc failed"; return w; }
    long off = 0;
    for (auto& n : names) {
        const RawTensor* t = raw(n);
        (void)hipMemcpyAsync(tmp + off, t->d, t->numel * 4, hipMemcpyDeviceToDevice, 0);
        off += t->numel;
    }
    int* dperm = nullptr;
    if (perm) {
        if (hipMalloc(&dperm, perm->size() * sizeof(int)) != hipSuccess) { (void)hipFree(tmp); return w; }
        (void)hipMemcpyAsync(dperm, perm->data(), perm->size() * sizeof(int), hipMemcpyHostToDevice, 0);
    }
    const int Nout = perm ? (int)perm->size() : Ntot;   // a permutation may also insert zero rows (entries -1): head padding
    w.N = Nout;
    w.K = K;
    w.hi = static_cast<bf16_t*>(dalloc((size_t)Nout * K * 2));
    w.lo = static_cast<bf16_t*>(dalloc((size_t)Nout * K * 2));
    w.h16 = static_cast<bf16_t*>(dalloc((size_t)Nout * K * 2));
    if (!w.hi || !w.lo || !w.h16 || launch_split_rows(tmp, K, w.hi, w.lo, K, Nout, K, dperm, 0, w.h16) != hipSuccess) w.N = 0;
    (void)hipStreamSynchronize(0);
    (void)hipFree(tmp);
    if (dperm) (void)hipFree(dperm);
    return w;
}

// concatenate 1-D tensors; a name "" inserts zero_len[k] zeros (k-th empty name)
float* Engine::concat_vec(const std::vector<std::string>& names, const std::vector<int>& zero_len) {
    long tot = 0;
    size_t zi = 0;
    for (auto& n : names) {
        if (n.empty()) { tot += zero_len[zi++]; continue; }
        const RawTensor* t = raw(n);
        if (!t) { err_ = "concat: missing tensor " + n; return nullptr; }
        tot += t->numel;
    }
    float* d = static_cast<float*>(dalloc(tot * 4));
    if (!d) return nullptr;
    (void)hipMemsetAsync(d, 0, tot * 4, 0);
    long off = 0;
    zi = 0;
    for (auto& n : names) {
        if (n.empty()) { off += zero_len[zi++]; continue; }
        const RawTensor* t = raw(n);
        (void)hipMemcpyAsync(d + off, t->d, t->numel * 4, hipMemcpyDeviceToDevice, 0);
        off += t->numel;
    }
    return d;
}

static std::vector<int> swiglu_perm(int F) {
    std::vector<int> p(2 * F);
    for (int i = 0; i < 2 * F; ++i) {
        int blk = i / 64, within = i % 64;
        p[i] = within < 32 ? blk * 32 + within : F + blk * 32 + (within - 32);
    }
    return p;
}

int Engine::build_encoder(EncoderW& e, const std::string& prefix, int dim, int heads, int ff, int layers, float eps) {
    e.dim = dim; e.heads = heads; e.dh = dim / heads; e.ff = ff; e.layers = layers; e.eps = eps;
    if (ff % 32) return fail("encoder ff must be a multiple of 32");
    std::vector<int> perm = swiglu_perm(ff);
    e.blocks.clear();
    for (int i = 0; i < layers; ++i) {
        std::string p = sidx(prefix + ".blocks.", i, "");
        for (const char* n : {"wq", "wk", "wv", "wo", "gate"})
            if (check_shape(p + ".attention." + n + ".weight", {dim, dim})) return 1;
        if (check_shape(p + ".attention.q_norm.weight", {heads, dim / heads}) || check_shape(p + ".attention.k_norm.weight", {heads, dim / heads}) ||
            check_shape(p + ".mlp.w1.weight", {ff, dim}) || check_shape(p + ".mlp.w3.weight", {ff, dim}) ||
            check_shape(p + ".mlp.w2.weight", {dim, ff}) || check_shape(p + ".attention_norm.weight", {dim}) ||
            check_shape(p + ".mlp_norm.weight", {dim}))
            return 1;
        EncBlockW b;
        b.qkvg = pack_rows({p + ".attention.wq.weight", p + ".attention.wk.weight", p + ".attention.wv.weight",
                            p + ".attention.gate.weight"});
        b.wo = pack_rows({p + ".attention.wo.weight"});
        b.ff13 = pack_rows({p + ".mlp.w1.weight", p + ".mlp.w3.weight"}, &perm);
        b.w2 = pack_rows({p + ".mlp.w2.weight"});
        b.qn = rawp(p + ".attention.q_norm.weight");
        b.kn = rawp(p + ".attention.k_norm.weight");
        b.an = rawp(p + ".attention_norm.weight");
        b.mn = rawp(p + ".mlp_norm.weight");
        if (!b.qkvg.N || !b.wo.N || !b.ff13.N || !b.w2.N || !b.qn || !b.kn || !b.an || !b.mn)
            return fail("encoder block incomplete: " + p + " (" + err_ + ")");
        e.blocks.push_back(b);
    }
    e.final_norm = rawp(prefix + ".norm.weight");
    if (!e.final_norm) return fail("missing " + prefix + ".norm.weight");
    return make_rope(e.dh, &e.rope_cos, &e.rope_sin);
}

int Engine::make_rope(int dim, float** cos_out, float** sin_out) {
    const size_t n = (size_t)kMaxPos * dim;
    float* ang = nullptr;
    HIPC(hipMalloc(&ang, n * 4));
    *cos_out = static_cast<float*>(dalloc(n * 4));
    *sin_out = static_cast<float*>(dalloc(n * 4));
    if (!*cos_out || !*sin_out) return fail("rope table alloc failed");
    HIPC(launch_rope_table(ang, kMaxPos, dim, 0));
    HIPC(launch_rope_cossin(ang, *cos_out, *sin_out, (int)n, 0));
    HIPC(hipStreamSynchronize(0));
    HIPC(hipFree(ang));
    return 0;
}

int Engine::finalize_dit() {
    const std::string T = "dit.transformer_blocks.";
    // shape contract of DiTModel(64).state_dict() (reference model.py:33-54): the kernels assume these strides, so a checkpoint
    // of another model size must fail here with the tensor's name instead of reading out of bounds later
    {
        const long H = kHidden, FF = kFF;
        struct { const char* n; std::initializer_list<long> sh; } g[] = {
            {"time_embedding.mlp.0.weight", {H, 256}}, {"time_embedding.mlp.0.bias", {H}},
            {"time_embedding.mlp.2.weight", {H, H}}, {"time_embedding.mlp.2.bias", {H}},
            {"dit.emb_proj.0.weight", {2 * H, H}}, {"dit.emb_proj.0.bias", {2 * H}},
            {"dit.emb_proj.2.weight", {H, 2 * H}}, {"dit.emb_proj.2.bias", {H}},
            {"dit.input_embed.proj.weight", {H, kLatent}}, {"dit.input_embed.proj.bias", {H}},
            {"dit.input_embed.conv_pos_embed.conv1.bias", {H}}, {"dit.input_embed.conv_pos_embed.conv2.bias", {H}},
            {"dit.phoneme_proj.weight", {H, 512}}, {"dit.phoneme_proj.bias", {H}},
            {"dit.norm_out.linear.weight", {2 * H, H}}, {"dit.norm_out.linear.bias", {2 * H}},
            {"velocity.weight", {kLatent, H}}, {"velocity.bias", {kLatent}},
            {"style_encoder.in_proj.weight", {512, kLatent}}, {"style_encoder.in_proj.bias", {512}},
            {"style_encoder.out_proj.weight", {H, 512}}, {"style_encoder.out_proj.bias", {H}},
            {"style_encoder.log_scale", {}}, {"phoneme_embedding.text_embedding.weight", {198, 512}},
        };
        for (auto& t : g)
            if (check_shape(t.n, t.sh)) return 1;
        for (int i = 0; i < kBlocks; ++i) {
            const std::string p = sidx(T, i, "");
            for (const char* n : {"to_q", "to_k_self", "to_v_self", "to_k_ref", "to_v_ref", "to_k_text", "to_v_text"})
                if (check_shape(p + ".attn." + n + ".weight", {H, H}) || check_shape(p + ".attn." + n + ".bias", {H})) return 1;
            if (check_shape(p + ".attn.gate.weight", {H, H}) || check_shape(p + ".attn.to_out.0.weight", {H, H})) return 1;
            for (const char* n : {"q_norm", "k_norm", "k_norm_cross"})
                if (check_shape(p + ".attn." + n + ".weight", {kHeads, kDh})) return 1;
            if (check_shape(p + ".attn_norm.linear.weight", {6 * H, H}) || check_shape(p + ".attn_norm.linear.bias", {6 * H})) return 1;
            if (check_shape(p + ".ff.w1.weight", {FF, H}) || check_shape(p + ".ff.w1.bias", {FF}) ||
                check_shape(p + ".ff.w3.weight", {FF, H}) || check_shape(p + ".ff.w3.bias", {FF}) ||
                check_shape(p + ".ff.w2.weight", {H, FF}) || check_shape(p + ".ff.w2.bias", {H}))
                return 1;
        }
    }
    time0_ = pack_rows({"time_embedding.mlp.0.weight"});
    time2_ = pack_rows({"time_embedding.mlp.2.weight"});
    emb0_ = pack_rows({"dit.emb_proj.0.weight"});
    emb2_ = pack_rows({"dit.emb_proj.2.weight"});
    inproj_ = pack_rows({"dit.input_embed.proj.weight"});
    velocity_ = pack_rows({"velocity.weight"});
    phproj_ = pack_rows({"dit.phoneme_proj.weight"});
    style_in_ = pack_rows({"style_encoder.in_proj.weight"});
    style_out_ = pack_rows({"style_encoder.out_proj.weight"});
    if (!time0_.N || !time2_.N || !emb0_.N || !emb2_.N || !inproj_.N || !velocity_.N || !phproj_.N || !style_in_.N ||
        !style_out_.N)
        return fail("DiT pack failed: " + err_);

    std::vector<std::string> modw, modb, kvr, kvrb, kvt, kvtb, knc;
    for (int i = 0; i < kBlocks; ++i) {
        std::string p = sidx(T, i, "");
        modw.push_back(p + ".attn_norm.linear.weight");
        modb.push_back(p + ".attn_norm.linear.bias");
        kvr.push_back(p + ".attn.to_k_ref.weight"); kvr.push_back(p + ".attn.to_v_ref.weight");
        kvrb.push_back(p + ".attn.to_k_ref.bias"); kvrb.push_back(p + ".attn.to_v_ref.bias");
        kvt.push_back(p + ".attn.to_k_text.weight"); kvt.push_back(p + ".attn.to_v_text.weight");
        kvtb.push_back(p + ".attn.to_k_text.bias"); kvtb.push_back(p + ".attn.to_v_text.bias");
        knc.push_back(p + ".attn.k_norm_cross.weight");
    }
    modw.push_back("dit.norm_out.linear.weight");
    modb.push_back("dit.norm_out.linear.bias");
    modall_ = pack_rows(modw);
    modall_b_ = concat_vec(modb);
    kvref_ = pack_rows(kvr);
    kvref_b_ = concat_vec(kvrb);
    kvtext_ = pack_rows(kvt);
    kvtext_b_ = concat_vec(kvtb);
    knc_ = concat_vec(knc);
    if (!modall_.N || !modall_b_ || !kvref_.N || !kvref_b_ || !kvtext_.N || !kvtext_b_ || !knc_)
        return fail("DiT modulation/KV pack failed: " + err_);
    if (modall_.N != kModLd) return fail("modulation table width mismatch");

    // grouped conv (dit.py:218-220) -> per-group GEMM weights [G*cpg][K*64] (ic padded 60 -> 64)
    for (int c = 0; c < 2; ++c) {
        const RawTensor* w = raw(sidx("dit.input_embed.conv_pos_embed.conv", c + 1, ".weight"));
        if (!w || w->shape.size() != 3 || w->shape[0] != kHidden || w->shape[1] != kConvCpg || w->shape[2] != kConvK)
            return fail("conv_pos_embed weight missing or wrong shape");
        float* tmp = nullptr;
        const int Kp = kConvK * kConvGs;
        HIPC(hipMalloc(&tmp, (size_t)kHidden * Kp * 4));
        GatherSpec g{0, 0, (long)kConvCpg * kConvK, 1, kConvK, kHidden, kConvGs, kConvCpg};
        HIPC(launch_gather_pack(w->d, tmp, kHidden, Kp, g, 0));
        (c ? conv2_ : conv1_) = pack_from_f32(tmp, kHidden, Kp);
        HIPC(hipStreamSynchronize(0));
        HIPC(hipFree(tmp));
        if (!(c ? conv2_ : conv1_).N) return fail("conv pack failed");
    }

    std::vector<int> perm = swiglu_perm(kFF);
    std::vector<int> hperm(4 * kHeads * 128, -1);   // row (part * 8 + h) * 128 + d <- source row part * 960 + h * 120 + d, pad rows zero
    for (int part = 0; part < 4; ++part)
        for (int h = 0; h < kHeads; ++h)
            for (int d = 0; d < kDh; ++d) hperm[(part * kHeads + h) * 128 + d] = part * kHidden + h * kDh + d;
    blocks_.clear();
    qkvg_unpadded_ready_ = false;
    for (int i = 0; i < kBlocks; ++i) {
        std::string p = sidx(T, i, "");
        DitBlockW b;
        b.name = p;
        if (!(attn_img_ && attn_epi_))   // (the default path reads the head-padded pack only: ensure_qkvg_unpadded builds this one on demand)
            b.qkvg = pack_rows({p + ".attn.to_q.weight", p + ".attn.to_k_self.weight", p + ".attn.to_v_self.weight",
                                p + ".attn.gate.weight"});
        b.b_qkvg = concat_vec({p + ".attn.to_q.bias", p + ".attn.to_k_self.bias", p + ".attn.to_v_self.bias", ""},
                              {kHidden});
        b.qkvgp = pack_rows({p + ".attn.to_q.weight", p + ".attn.to_k_self.weight", p + ".attn.to_v_self.weight",
                             p + ".attn.gate.weight"}, &hperm);
        b.b_qkvgp = static_cast<float*>(dalloc(hperm.size() * 4));
        if (b.b_qkvgp && b.b_qkvg) {   // the padded bias: gather on the host side of a tiny vector
            std::vector<float> src(4 * kHidden), dst(hperm.size(), 0.f);
            (void)hipMemcpy(src.data(), b.b_qkvg, src.size() * 4, hipMemcpyDeviceToHost);
            for (size_t r = 0; r < hperm.size(); ++r) if (hperm[r] >= 0) dst[r] = src[hperm[r]];
            (void)hipMemcpy(b.b_qkvgp, dst.data(), dst.size() * 4, hipMemcpyHostToDevice);
        }
        b.out = pack_rows({p + ".attn.to_out.0.weight"});
        b.ff13 = pack_rows({p + ".ff.w1.weight", p + ".ff.w3.weight"}, &perm);
        b.ff2 = pack_rows({p + ".ff.w2.weight"}, nullptr, kFFp);
        b.b1 = rawp(p + ".ff.w1.bias"); b.b3 = rawp(p + ".ff.w3.bias"); b.b2 = rawp(p + ".ff.w2.bias");
        b.qn = rawp(p + ".attn.q_norm.weight"); b.kn = rawp(p + ".attn.k_norm.weight");
        if ((!(attn_img_ && attn_epi_) && !b.qkvg.N) || !b.qkvgp.N || !b.b_qkvg || !b.b_qkvgp || !b.out.N || !b.ff13.N || !b.ff2.N || !b.b1 || !b.b3 || !b.b2 || !b.qn || !b.kn)
            return fail("DiT block incomplete: " + p + " (" + err_ + ")");
        blocks_.push_back(b);
    }
    if (build_encoder(style_, "style_encoder", 512, 8, 1536, 12, 1e-5f)) return 1;
    if (build_encoder(text_, "phoneme_embedding", 512, 4, 1024, 8, 1e-6f)) return 1;
    float ls = 0.f;
    if (!raw("style_encoder.log_scale")) return fail("missing style_encoder.log_scale");
    HIPC(hipMemcpy(&ls, rawp("style_encoder.log_scale"), 4, hipMemcpyDeviceToHost));
    style_scale_ = expf(ls);
    if (make_rope(64, &rope_dit_cos_, &rope_dit_sin_)) return 1;
    for (const char* n : {"time_embedding.mlp.0.bias", "time_embedding.mlp.2.bias", "dit.emb_proj.0.bias",
                          "dit.emb_proj.2.bias", "dit.input_embed.proj.bias", "velocity.bias", "dit.phoneme_proj.bias",
                          "style_encoder.in_proj.bias", "style_encoder.out_proj.bias",
                          "dit.input_embed.conv_pos_embed.conv1.bias", "dit.input_embed.conv_pos_embed.conv2.bias",
                          "phoneme_embedding.text_embedding.weight"})
        if (!raw(n)) return fail(std::string("missing tensor ") + n);
    dit_ready_ = true;
    return 0;
}

int Engine::finalize_codec(bool decoder) {
    const CodecSpecC& s = cspec_;
    CodecHalfW& h = decoder ? dec_ : enc_;
    const std::string P = decoder ? "codec.decoder" : "codec.encoder";
    const int S = s.n_ratios + 1, Kc = s.kernel;
    if (Kc - 1 > kCodecPad) return fail("codec kernel too large for the frame padding");
    h = CodecHalfW();
    // Optional tensors (CodecSpec flags conv_bias / ffn_bias / layer_scale, smalltts_amd/weights.py): an exported codec may come
    // without biases or layer scales — absent means the identity value, present means the inventory's shape.
    auto opt_vec = [&](const std::string& name, long n, float fill, const float** out) -> int {
        if (raw(name)) {
            if (check_shape(name, {n})) return 1;
            *out = rawp(name);
            return 0;
        }
        float* d = static_cast<float*>(dalloc((size_t)n * 4));
        if (!d) return fail("codec: out of memory for the default of " + name);
        HIPC(launch_fill(d, fill, n, 0));
        *out = d;
        return 0;
    };
    auto gather_to_pw = [&](const float* src, int N, int K, GatherSpec g, PW& out, float** f32_out, bool x2 = false) -> int {
        float* tmp = nullptr;
        if (f32_out) {
            tmp = static_cast<float*>(dalloc((size_t)N * K * 4));
            if (!tmp) return fail("codec pack alloc failed");
        } else {
            HIPC(hipMalloc(&tmp, (size_t)N * K * 4));
        }
        HIPC(launch_gather_pack(src, tmp, N, K, g, 0));
        if (f32_out) { *f32_out = tmp; return 0; }
        out = pack_from_f32(tmp, N, K);
        if (x2 && out.N) {   // low half of the fp16 pair (PREC_F16X2)
            out.l16 = static_cast<bf16_t*>(dalloc((size_t)N * K * 2));
            if (!out.l16) return fail("codec pack alloc failed");
            HIPC(launch_f16_residual(tmp, K, out.h16, out.l16, out.K, N, K, 0));
        }
        HIPC(hipStreamSynchronize(0));
        HIPC(hipFree(tmp));
        return out.N ? 0 : fail("codec pack failed");
    };
    for (int i = 0; i < S; ++i) {
        CodecStageW st;
        const int C = decoder ? s.n_filters << (S - 1 - i) : s.n_filters << i;
        st.C = C;
        if (i > 0) {
            const int r = decoder ? s.ratios[i - 1] : s.ratios[s.n_ratios - i];
            if (r > kCodecPad) return fail("codec ratio exceeds frame padding");
            st.r = r;
            if (decoder) {
                const int Cin = 2 * C;
                const RawTensor* w = raw(sidx(P + ".up.", i, ".weight"));
                if (!w || w->numel != (long)Cin * C * 2 * r) return fail("missing/mis-shaped " + sidx(P + ".up.", i, ".weight"));
                GatherSpec g{r, 1, 2L * r, -(long)r, (long)C * 2 * r, C, Cin, Cin};
                if (gather_to_pw(w->d, r * C, 2 * Cin, g, st.resample, nullptr, true)) return 1;
                const float* b = nullptr;
                if (opt_vec(sidx(P + ".up.", i, ".bias"), C, 0.f, &b)) return 1;
                GatherSpec gb{0, 0, 0, 0, 1, 1, C, C};
                PW dummy;
                if (gather_to_pw(b, 1, r * C, gb, dummy, &st.resample_bias)) return 1;
            } else {
                const int Cin = C / 2;
                const RawTensor* w = raw(sidx(P + ".down.", i, ".weight"));
                if (!w || w->numel != (long)C * Cin * 2 * r) return fail("missing/mis-shaped " + sidx(P + ".down.", i, ".weight"));
                GatherSpec g{0, 0, (long)Cin * 2 * r, 1, 2L * r, C, Cin, Cin};
                if (gather_to_pw(w->d, C, 2 * r * Cin, g, st.resample, nullptr)) return 1;
                const float* b = nullptr;
                if (opt_vec(sidx(P + ".down.", i, ".bias"), C, 0.f, &b)) return 1;
                st.resample_bias = const_cast<float*>(b);
            }
        }
        const int depth = decoder ? s.depths[i] : s.depths[S - 1 - i];
        for (int j = 0; j < depth; ++j) {
            std::string p = P + ".stages." + std::to_string(i) + "." + std::to_string(j);
            CodecBlockW b;
            b.norm_w = rawp(p + ".norm.weight");
            b.ffn_norm_w = rawp(p + ".ffn_norm.weight");
            const float* mw = rawp(p + ".mixer.weight");
            {
                const long Fh = (long)s.ffn_mult * C;
                if (check_shape(p + ".norm.weight", {C}) || check_shape(p + ".mixer.weight", {C, Kc}) ||
                    check_shape(p + ".ffn_norm.weight", {C}) || check_shape(p + ".ffn.w1.weight", {Fh, C}) ||
                    check_shape(p + ".ffn.w2.weight", {C, Fh}))
                    return 1;
                if (opt_vec(p + ".mixer.bias", C, 0.f, &b.dw_b) || opt_vec(p + ".gamma", C, 1.f, &b.gamma) ||
                    opt_vec(p + ".ffn.w1.bias", Fh, 0.f, &b.b1) || opt_vec(p + ".ffn.w2.bias", C, 0.f, &b.b2) ||
                    opt_vec(p + ".ffn_gamma", C, 1.f, &b.ffn_gamma))
                    return 1;
            }
            GatherSpec g{0, 0, 1, 0, Kc, Kc, C, C};
            PW dummy;
            if (gather_to_pw(mw, Kc, C, g, dummy, &b.dw_w)) return 1;
            b.w1 = pack_rows({p + ".ffn.w1.weight"});
            b.w2 = pack_rows({p + ".ffn.w2.weight"});
            if (!b.w1.N || !b.w2.N) return fail("codec ffn pack failed: " + p + " " + err_);
            const int F = s.ffn_mult * C;
            if ((C == 128 || C == 256) && F == 4 * C && b.w1.K == C && b.w2.K == F) {  // streamed fused FFN: W2 hidden-tile-major
                b.w2t.hi = static_cast<bf16_t*>(dalloc((size_t)C * F * 2));
                b.w2t.lo = static_cast<bf16_t*>(dalloc((size_t)C * F * 2));
                b.w2t.h16 = static_cast<bf16_t*>(dalloc((size_t)C * F * 2));
                if (!b.w2t.hi || !b.w2t.lo || !b.w2t.h16) return fail("codec w2 tile pack: out of memory");
                HIPC(launch_w2_tile_pack(b.w2.hi, b.w2t.hi, C, F, 0));
                HIPC(launch_w2_tile_pack(b.w2.lo, b.w2t.lo, C, F, 0));
                HIPC(launch_w2_tile_pack(b.w2.h16, b.w2t.h16, C, F, 0));
                b.w2t.N = (F / 32) * C;
                b.w2t.K = 32;
            }
            if (certify_codec_ffn(p, b, C, F)) return 1;
            st.blocks.push_back(b);
        }
        h.stages.push_back(st);
    }
    const int C0 = h.stages[0].C, Cl = h.stages[S - 1].C;
    if (decoder) {
        const RawTensor* w = raw(P + ".stem.weight");
        if (!w || w->numel != (long)C0 * s.latent_dim * Kc) return fail("missing decoder stem");
        GatherSpec g{0, 0, (long)s.latent_dim * Kc, 1, Kc, C0, s.latent_dim, s.latent_dim};
        if (gather_to_pw(w->d, C0, Kc * s.latent_dim, g, h.stem, nullptr)) return 1;
        if (opt_vec(P + ".stem.bias", C0, 0.f, &h.stem_b)) return 1;
        const float* hw = rawp(P + ".head.weight");
        const float* hb = nullptr;
        if (opt_vec(P + ".head.bias", 1, 0.f, &hb)) return 1;
        if (!hw) return fail("missing decoder head");
        if (check_shape(P + ".head.weight", {1, Cl, Kc})) return 1;
        GatherSpec gh{0, 0, 1, 0, Kc, Kc, Cl, Cl};
        PW dummy;
        if (gather_to_pw(hw, Kc, Cl, gh, dummy, &h.head_w)) return 1;
        HIPC(hipMemcpy(&h.head_b_host, hb, 4, hipMemcpyDeviceToHost));
    } else {
        h.stem_w_raw = rawp(P + ".stem.weight");
        const RawTensor* w = raw(P + ".head.weight");
        if (opt_vec(P + ".stem.bias", C0, 0.f, &h.stem_b) || opt_vec(P + ".head.bias", s.latent_dim, 0.f, &h.head_b)) return 1;
        if (!h.stem_w_raw || !w) return fail("missing encoder stem/head");
        if (check_shape(P + ".stem.weight", {C0, 1, Kc}) || check_shape(P + ".head.weight", {s.latent_dim, Cl, Kc})) return 1;
        GatherSpec g{0, 0, (long)Cl * Kc, 1, Kc, s.latent_dim, Cl, Cl};
        if (gather_to_pw(w->d, s.latent_dim, Kc * Cl, g, h.head, nullptr)) return 1;
    }
    // optional RMSNorm in front of the head conv (CodecSpec.final_norm)
    h.final_norm_w = nullptr;
    if (raw(P + ".final_norm.weight")) {
        if (check_shape(P + ".final_norm.weight", {Cl})) return 1;
        h.final_norm_w = rawp(P + ".final_norm.weight");
    }
    h.ready = true;
    return 0;
}

// The fused codec FFN kernels (codec_ffn_wave / codec_ffn_stream: C <= 256) convert their normalised input and their GELU
// hidden to fp16 in registers and are VALU-bound: a per-value range check there would cost every batch ~4 % of the decoder for
// an event the weights can RULE OUT.  The input is RMS-normalised, so both are bounded by the weights alone
// (ffn_range_bound_kernel); a block whose bound stays inside the fp16 range needs no run-time check, one whose bound does not
// is reported through get_saturations (static part of SITE_CODEC_FFN), which makes the host side demote the site.
static constexpr float kFfnCertLimit = 65504.f / 1.002f;
int Engine::certify_codec_ffn(const std::string& name, CodecBlockW& b, int C, int F) {
    if (!(C == 32 || C == 64 || C == 128 || C == 256) || F != 4 * C) return 0;   // the wider stages' hiddens are counted at run time
    if (!cert_scratch_) {   // no scratch: nothing can be certified — the block runs split-bf16, and that is REPORTED
        b.f16_ok = false;
        ++sat_static_[SITE_CODEC_FFN];
        range_report_ += name + ": fused FFN range not certified (no certificate scratch); ";
        return 0;
    }
    const float* w1 = rawp(name + ".ffn.w1.weight");
    if (!w1) return 0;
    HIPC(hipMemsetAsync(cert_scratch_, 0, 2 * sizeof(float), 0));
    HIPC(launch_ffn_range_bound(w1, b.b1, b.ffn_norm_w, F, C, cert_scratch_, 0));
    float bound[2] = {0.f, 0.f};
    HIPC(hipMemcpy(bound, cert_scratch_, sizeof bound, hipMemcpyDeviceToHost));
    const float worst = bound[0] > bound[1] ? bound[0] : bound[1];
    if (worst > range_worst_) range_worst_ = worst;
    // The bound is evaluated on the fp32 weights; the kernels multiply fp16-rounded W1 rows with fp16-rounded inputs (2^-11 relative
    // each), so the hidden they convert can exceed it by ~(1 + 2^-11)^2 = 1.001.  Certified blocks convert WITHOUT a clamp (an
    // inf there turns the packed GELU into NaN), hence a margin instead of the bare fp16 maximum.
    if (!(worst <= kFfnCertLimit)) {
        b.f16_ok = false;
        ++sat_static_[SITE_CODEC_FFN];
        char buf[256];
        snprintf(buf, sizeof buf, "%s: fused FFN hidden bound %.4g, input bound %.4g exceed the fp16 range; ", name.c_str(), bound[0], bound[1]);
        range_report_ += buf;
    }
    return 0;
}

int Engine::finalize() {
    HIPC(hipSetDevice(device_));
    HIPC(hipDeviceSynchronize());  // nothing in flight may still read the packs of an earlier finalize()
    invalidate();
    free_packs();
    for (unsigned& v : sat_static_) v = 0;
    range_report_.clear();
    range_worst_ = 0.f;
    struct PackMode { bool& f; explicit PackMode(bool& b) : f(b) { f = true; } ~PackMode() { f = false; } } pm(packing_);
    if (raw("velocity.weight")) {
        if (finalize_dit()) return 1;
    }
    if (raw("codec.decoder.stem.weight")) {
        if (finalize_codec(true)) return 1;
    }
    if (raw("codec.encoder.stem.weight")) {
        if (finalize_codec(false)) return 1;
    }
    HIPC(hipDeviceSynchronize());
    finalized_ = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// small helpers to build GEMM calls
// ---------------------------------------------------------------------------------------------
static inline GemmOperands ops(const float* A, RowMap amap, const PW& w, int M, int row0 = 0, int nrows = -1) {
    GemmOperands g;
    g.A = A;
    g.amap = amap;
    g.Whi = w.hi + (long)row0 * w.K;
    g.Wlo = w.lo + (long)row0 * w.K;
    g.ldw = w.K;
    g.M = M;
    g.N = nrows < 0 ? w.N : nrows;
    g.K = w.K;
    g.a_z = 0;
    g.w_z = 0;
    g.w_zmod = 0;
    return g;
}
struct SplitBuf {  // activation buffer feeding gemm3 as its A operand: a split bf16 pair (x ~= hi + lo), or — for a consumer
    bf16_t* hi;     // of precision PREC_F16 / PREC_BF16 — one 16-bit array in `hi` (`lo` allocated but unused)
    bf16_t* lo;
    // the (hi, lo) pair a PRODUCER is handed so that it writes the format a consumer of precision `prec` reads (common.hpp)
    // (`sat`: the consumer site's saturation counter, counted into by fp16 producers — common.hpp sat_note)
    SplitBuf as(int prec, unsigned* sat = nullptr) const { return SplitBuf{hi, sm_lo_for(prec, lo, sat)}; }
};
static inline Gemm3Operands ops3(SplitBuf a, RowMap amap, const PW& w, int M, int prec, int row0 = 0, int nrows = -1) {
    Gemm3Operands g;
    g.Ahi = a.hi;
    g.Alo = prec == PREC_BF16X3 ? a.lo : nullptr;
    g.amap = amap;
    g.Whi = (prec == PREC_F16 ? w.h16 : w.hi) + (long)row0 * w.K;
    g.Wlo = prec == PREC_BF16X3 ? w.lo + (long)row0 * w.K : nullptr;
    g.ldw = w.K;
    g.M = M;
    g.N = nrows < 0 ? w.N : nrows;
    g.K = w.K;
    g.a_z = 0;
    g.w_z = 0;
    g.w_zmod = 0;
    g.ksplit_tiles = 0;
    return g;
}
static inline EpiStore<ACT_NONE> store_to(float* out, RowMap omap, const float* bias, float scale = 1.f,
                                          const uint8_t* rowmask = nullptr) {
    return EpiStore<ACT_NONE>{out, omap, 0, bias, 0, scale, rowmask, nullptr, nullptr};
}
static inline EpiStore<ACT_NONE> store_split_to(SplitBuf o, RowMap omap, const float* bias, float scale = 1.f,
                                                const uint8_t* rowmask = nullptr) {
    return EpiStore<ACT_NONE>{nullptr, omap, 0, bias, 0, scale, rowmask, o.hi, o.lo};
}
template <class T>
static inline SplitBuf take_split(T& bump, size_t elems) {
    SplitBuf s;
    s.hi = bump.template take<bf16_t>(elems);
    s.lo = bump.template take<bf16_t>(elems);
    return s;
}

// Split-K residual GEMM for the small-M DiT projections (N = 960 gives only 150 tiles): the K range is cut into
// `splits` slices (grid.z), each writes an fp32 partial, and one fused kernel reduces them in a fixed order and
// applies bias / tanh-gate / row mask / residual add -> deterministic, unlike atomics.
struct NextLN {  // optional: the AdaLN that follows the residual, fused into the split-K reduction kernel
    const float* shift = nullptr;
    const float* scale = nullptr;
    bf16_t* yhi = nullptr;
    bf16_t* ylo = nullptr;
    bool rms = false;   // true: RMSNorm with weight `shift` (encoders) instead of LayerNorm * (1 + scale) + shift
    float eps = 1e-6f;
};
static hipError_t gemm3_resid_splitk(const Gemm3Operands& g0, const EpiResid<0>& r, float* partial, int splits, int split,
                                     hipStream_t st, const NextLN& ln = NextLN(), int cfg = G3_64x64) {
    Gemm3Operands g = g0;
    const int nk = g.K / 64;
    g.ksplit_tiles = (nk + splits - 1) / splits;
    const int used = (nk + g.ksplit_tiles - 1) / g.ksplit_tiles;
    EpiStore<ACT_NONE> e{partial, rowmap_plain(g.N), (long)g.M * g.N, nullptr, 0, 1.f, nullptr, nullptr, nullptr};
    hipError_t err = gemm3_store(g, ACT_NONE, e, used, split, st, cfg);
    if (err != hipSuccess) return err;
    if (ln.shift)
        return launch_splitk_resid_ln(partial, used, r.x, r.bias, r.gate, r.gld, r.grow0, r.grstride, r.rows_per_batch,
                                      r.rowmask, g.M, g.N, ln.eps, ln.shift, ln.scale, ln.yhi, ln.ylo, st, ln.rms);
    return launch_splitk_resid(partial, used, r.x, r.bias, r.gate, r.gld, r.grow0, r.grstride, r.rows_per_batch, r.rowmask,
                               g.M, g.N, st, &r.xmap);
}

// K slices that spread a weight-bound product (few rows, long K) over the chip.  The count depends on (K, row class) only, never
// on M itself, so that results stay bit-identical across batch sizes inside a class (tests/test_fullsize_gpu.py: a batch of 2
// shares its prefix with the batch of 1).  Slices of >= 16 k-tiles (>= 32 in the 513..2048-row class).  1 = leave the product alone.
static int small_m_splits(int M, int K, int max_few, int max_some) {
    if (!g_small_m_splitk) return 1;
    const int nk = K / 64;
    int sp = 1;
    if (M <= 512) sp = nk / 16 < max_few ? nk / 16 : max_few;
    else if (M <= 2048) sp = nk / 32 < max_some ? nk / 32 : max_some;
    return sp < 2 ? 1 : sp;
}
// out = A x W^T + bias as K slices into fp32 partials + one fixed-order reduce pass (no residual: the reduce overwrites)
static hipError_t gemm3_store_splitk(const Gemm3Operands& g0, float* partial, int splits, int split, const float* bias, float* out,
                                     const RowMap& omap, hipStream_t st) {
    Gemm3Operands g = g0;
    const int nk = g.K / 64;
    g.ksplit_tiles = (nk + splits - 1) / splits;
    const int used = (nk + g.ksplit_tiles - 1) / g.ksplit_tiles;
    EpiStore<ACT_NONE> e{partial, rowmap_plain(g.N), (long)g.M * g.N, nullptr, 0, 1.f, nullptr, nullptr, nullptr};
    hipError_t err = gemm3_store(g, ACT_NONE, e, used, split, st, G3_64x64);
    if (err != hipSuccess) return err;
    return launch_splitk_resid(partial, used, out, bias, nullptr, 0, 0, 0, 1, nullptr, g.M, g.N, st, &omap, true);
}

// ---------------------------------------------------------------------------------------------
// K10: encoder stack (style.py:70-105 / phonemes.py:131-167), x (fp32 residual) updated in place.
// GEMM inputs (y, o, ffh) live as split bf16 pairs written by the producing kernel.
// ---------------------------------------------------------------------------------------------
namespace {
struct EncWs {
    float *x, *qkvg, *seq, *part, *lnpart;
    SplitBuf y, o, ffh, seqs;
    SplitBuf qi, ki, vti, gi;   // attention operand images (attention_img.hip): [B][H][S][dhp] x 2, [B][H][dhp][pad8(S)], [M][D]
    size_t vt_elems;
    void plan(Bump& b, int B, int S) {
        const int Mx = B * S > 0 ? B * S : 1;
        x = b.take<float>((size_t)Mx * 512);
        part = b.take<float>((size_t)kSplitK * Mx * 512);
        lnpart = b.take<float>((size_t)Mx * (512 / 32) * 2);
        qkvg = b.take<float>((size_t)Mx * 2048);
        seq = b.take<float>((size_t)Mx * kHidden);
        y = take_split(b, (size_t)Mx * 512);
        o = take_split(b, (size_t)Mx * 512);
        ffh = take_split(b, (size_t)Mx * 1536);
        seqs = take_split(b, (size_t)Mx * kHidden);
        // H * dhp = 512 for both encoders (8 x 64, 4 x 128); V^T rows are padded to 8 keys per utterance: B x 512 x pad8(S)
        // (round 3 took 8 Mx x 512 here, "at most Mx + 7 B <= 8 Mx": 16 KB per row, GBs at large B x P — ADVICE r3)
        qi = take_split(b, (size_t)Mx * 512);
        ki = take_split(b, (size_t)Mx * 512);
        vt_elems = (size_t)(B > 0 ? B : 1) * 512 * (size_t)pad8(S > 0 ? S : 1);
        vti = take_split(b, vt_elems);
        gi = take_split(b, (size_t)Mx * 512);
    }
};
}  // namespace

int Engine::run_encoder(hipStream_t st, const EncoderW& e, void* wsv, int B, int S, const uint8_t* key_mask) {
    // On return w.y holds RMSNorm(x; final_norm) as a split pair: every norm except the first is fused into the
    // split-K reduction of the residual GEMM in front of it (the tiny-M projections get 4x the workgroups that way).
    EncWs& w = *static_cast<EncWs*>(wsv);
    const int M = B * S, D = e.dim;
    const RowMap rd = rowmap_plain(D);
    if (e.blocks.empty()) return fail("encoder without blocks");
    const int pe = prec_[SITE_ENCODER];
    unsigned* const se = satp(SITE_ENCODER);
    const SplitBuf y = w.y.as(pe, se), o = w.o.as(pe, se), ffh = w.ffh.as(pe, se);  // every activation here feeds a SITE_ENCODER GEMM
    HIPC(launch_rmsnorm(w.x, rd, nullptr, y.hi, y.lo, rd, M, D, e.eps, e.blocks[0].an, st));
    // RMSNorm fold (gemm.hpp LnFoldIn, rms): the norm between two block GEMMs lives in their epilogues — the producer writes x w and the
    // row's sum-of-squares partials, the consumer scales by rstd; the first norm (above) and the final one (feeds a plain projection) stay
    const bool fold = ln_fold_now() && attn_img_ && attn_epi_ && D % 32 == 0;
    LnFoldIn fin;
    fin.part = w.lnpart; fin.NP = D / 32; fin.inv_c = 1.0f / D; fin.eps = e.eps; fin.rms = 1;
    for (size_t l = 0; l < e.blocks.size(); ++l) {
        const EncBlockW& b = e.blocks[l];
        const bool epi = attn_img_ && attn_epi_;   // the GEMM's own epilogue writes the attention operands
        if (!epi) HIPC(gemm3_store(ops3(w.y, rd, b.qkvg, M, pe), ACT_NONE, store_to(w.qkvg, rowmap_plain(4 * D), nullptr), 1, pe, st));
        AttnArgs a{};
        a.q = w.qkvg; a.k = w.qkvg + D; a.v = w.qkvg + 2 * D; a.gate = w.qkvg + 3 * D;
        a.bs = (long)S * 4 * D; a.rs = 4 * D;
        a.qw = b.qn; a.kw = b.kn; a.eps = e.eps;
        a.rope_cos = e.rope_cos; a.rope_sin = e.rope_sin; a.rot_dim = e.dh;
        a.mask_self = key_mask;
        a.out = nullptr; a.out_hi = o.hi; a.out_lo = o.lo; a.obs = (long)S * D; a.ors = D;
        a.B = B; a.N = S; a.H = e.heads; a.dh = e.dh;
        if (attn_img_) {
            // q / k head-norm + RoPE + operand formatting by the producer, then a DMA + MFMA attention kernel (attention_img.hip)
            const int pa = prec_[SITE_ATTN], Sp = pad8(S);
            QkvPackArgs pk{};
            pk.qkvg = w.qkvg; pk.qw = b.qn; pk.kw = b.kn; pk.eps = e.eps; pk.q_scale = 1.0f / sqrtf((float)e.dh);
            pk.rope_cos = e.rope_cos; pk.rope_sin = e.rope_sin; pk.rot_dim = e.dh; pk.prec = pa;
            pk.q = w.qi.hi; pk.q_lo = img_lo(pa, w.qi.lo); pk.k = w.ki.hi; pk.k_lo = img_lo(pa, w.ki.lo); pk.vt = w.vti.hi; pk.vt_lo = img_lo(pa, w.vti.lo);
            pk.g = w.gi.hi; pk.g_lo = img_lo(pa, w.gi.lo);
            pk.B = B; pk.N = S; pk.H = e.heads; pk.dh = e.dh; pk.dhp = e.dh <= 64 ? 64 : 128; pk.Np = Sp;
            if (l == 0 && Sp != S) {   // pad key columns of V^T: zero once per call (the producer only writes n < S)
                HIPC(hipMemsetAsync(w.vti.hi, 0, (size_t)B * e.heads * pk.dhp * Sp * 2, st));
                if (pa == PREC_BF16X3) HIPC(hipMemsetAsync(w.vti.lo, 0, (size_t)B * e.heads * pk.dhp * Sp * 2, st));
            }
            if (epi) {
                EpiQKV eq{nullptr, pk.qw, pk.kw, pk.rope_cos, pk.rope_sin, pk.eps, pk.q_scale, pk.rot_dim, pa,
                          pk.q, pk.q_lo, pk.k, pk.k_lo, pk.vt, pk.vt_lo, pk.g, pk.g_lo, S, e.heads, e.dh, pk.dhp, Sp};
                if (fold && l > 0) eq.fold = fin;   // w.y = x attention_norm.weight, written by the previous block's FF2 epilogue
                HIPC(gemm3_qkv(ops3(w.y, rd, b.qkvg, M, pe), eq, pe, st));   // (both encoders' heads are 64 / 128 wide: no padding)
            } else {
                HIPC(launch_qkv_pack(pk, st));
            }
            AttnImg ai{};
            ai.prec = pa;
            ai.q = pk.q; ai.q_lo = w.qi.lo; ai.k = pk.k; ai.k_lo = w.ki.lo; ai.vt = pk.vt; ai.vt_lo = w.vti.lo;   // (the consumer reads the lo ARRAYS, split format only)
            ai.g = pk.g; ai.g_lo = sm_lo_for(pa, w.gi.lo);
            ai.mask_self = key_mask;
            ai.out_hi = o.hi; ai.out_lo = o.lo; ai.ors = D;
            ai.B = B; ai.N = S; ai.H = e.heads; ai.dh = e.dh; ai.Np = Sp;
            HIPC(launch_attention_img(ai, st));
        } else {
#ifdef SMTTS_TEST_KERNELS   // the fp32 VALU reference attention (attention.hip): test builds only
            a.prenormed = 1;
            HIPC(launch_qk_prep(a, st));
            HIPC(launch_attention(a, st));
#else
            return fail("the fp32 VALU reference attention is not part of this build (make TEST_KERNELS=1)");
#endif
        }
        EpiResid<0> r1{w.x, rd, nullptr, nullptr, 0, 0, 0, 1, nullptr};
        NextLN n1{b.mn, nullptr, y.hi, y.lo, true, e.eps};
        if (fold) {
            EpiResidLN e1{w.x, rd, nullptr, nullptr, nullptr, b.mn, y.hi, y.lo, D, w.lnpart, D / 32, 1};
            HIPC(gemm3_resid_ln(ops3(w.o, rd, b.wo, M, pe), e1, pe, st));
        } else if (ksplit_enc_ > 1) {
            HIPC(gemm3_resid_splitk(ops3(w.o, rd, b.wo, M, pe), r1, w.part, ksplit_enc_, pe, st, n1));
        } else {
            HIPC(gemm3_resid(ops3(w.o, rd, b.wo, M, pe), 0, r1, pe, st));
            HIPC(launch_rmsnorm(w.x, rd, nullptr, y.hi, y.lo, rd, M, D, e.eps, n1.shift, st));
        }
        EpiSwiGLU sw{nullptr, e.ff, nullptr, nullptr, ffh.hi, ffh.lo};
        if (fold) sw.fold = fin;
        HIPC(gemm3_swiglu(ops3(w.y, rd, b.ff13, M, pe), sw, pe, st));
        NextLN n2{l + 1 < e.blocks.size() ? e.blocks[l + 1].an : e.final_norm, nullptr, y.hi, y.lo, true, e.eps};
        if (fold && l + 1 < e.blocks.size()) {
            EpiResidLN e2{w.x, rd, nullptr, nullptr, nullptr, e.blocks[l + 1].an, y.hi, y.lo, D, w.lnpart, D / 32, 1};
            HIPC(gemm3_resid_ln(ops3(w.ffh, rowmap_plain(e.ff), b.w2, M, pe), e2, pe, st));
        } else if (ksplit_enc_ > 1) {
            HIPC(gemm3_resid_splitk(ops3(w.ffh, rowmap_plain(e.ff), b.w2, M, pe), r1, w.part, ksplit_enc_, pe, st, n2));
        } else {
            HIPC(gemm3_resid(ops3(w.ffh, rowmap_plain(e.ff), b.w2, M, pe), 0, r1, pe, st));
            HIPC(launch_rmsnorm(w.x, rd, nullptr, y.hi, y.lo, rd, M, D, e.eps, n2.shift, st));
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// E0: condition encoder (model.py:88-95)
// ---------------------------------------------------------------------------------------------
size_t Engine::cond_ws_bytes(int B, int R, int P) const {
    Bump b(nullptr);
    EncWs ws, wt;  // the style and the text encoder run concurrently: one workspace each
    ws.plan(b, B, R);
    wt.plan(b, B, P);
    return b.off + 256;
}

// Side stream for the text-encoder half of cond_encode: forked from / joined into the caller's stream with
// events, so the caller still sees a single-stream operator.  Both encoders are tiny-M launch chains (120 / 240
// rows) that each fill a fraction of the chip; run side by side they overlap almost completely.
// One side stream + event pair PER CALLER STREAM (round 6): a caller that keeps several batches in flight gives each its own stream, and
// each of those forks onto its own side stream — the text encoders / modulation chains of batches in flight no longer queue behind one
// another on a single engine-wide stream.  The set of the call being enqueued is installed in aux_ / ev_fork_ / ev_join_ (one host thread
// enqueues one call at a time).
int Engine::ensure_aux(hipStream_t st) {
    AuxSet& a = aux_sets_[st];
    if (!a.stream) {
        if (aux_sets_.size() > 64) return fail("side streams: more than 64 caller streams on one engine");
        HIPC(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
        HIPC(hipEventCreateWithFlags(&a.fork, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&a.join, hipEventDisableTiming));
    }
    aux_ = a.stream; ev_fork_ = a.fork; ev_join_ = a.join;
    return 0;
}

int Engine::cond_encode(hipStream_t st, const float* ref, const int64_t* ref_len, const int64_t* ids,
                        const uint8_t* ph_mask, int B, int R, int P, float* k_ref, float* v_ref, uint8_t* ref_mask,
                        float* k_text, float* v_text, void* ws, size_t ws_bytes, float* ref_seq_out, float* mem_out) {
    DeepScope deep_scope(gemm_deep_);
    ProfTag ptag("enc");
    if (!dit_ready_) return fail("cond_encode: DiT weights not finalized");
    if (R > kMaxPos || P > kMaxPos) return fail("cond_encode: sequence longer than the rope table (4096)");
    if (ws_bytes < cond_ws_bytes(B, R, P)) return fail("cond_encode: workspace too small");
    HIPC(hipSetDevice(device_));
    Bump bump(ws);
    EncWs w, wt;
    w.plan(bump, B, R);
    wt.plan(bump, B, P);
    const RowMap r512 = rowmap_plain(512), rh = rowmap_plain(kHidden);
    const int pe = prec_[SITE_ENCODER], pk = prec_[SITE_CROSS_KV], pc = prec_[SITE_COND];
    const bool fork = dual_stream_ && R > 0 && P > 0;
    hipStream_t stt = st;  // stream of the text half
    if (fork) {
        if (ensure_aux(st)) return 1;
        HIPC(hipEventRecord(ev_fork_, st));  // inputs produced on the caller's stream are visible to the side stream
        HIPC(hipStreamWaitEvent(aux_, ev_fork_, 0));
        stt = aux_;
    }

    // ---- E2 text encoder (phonemes.py:200-207) + phoneme_proj (dit.py:293-298) ---------------
    if (P > 0) {
        const int M = B * P;
        HIPC(launch_embedding(ids, rawp("phoneme_embedding.text_embedding.weight"), wt.x, M, 512, 198, stt));
        if (run_encoder(stt, text_, &wt, B, P, ph_mask)) return 1;  // leaves RMSNorm(x; final_norm) in wt.y
        float* mem = mem_out ? mem_out : wt.seq;
        HIPC(gemm3_store(ops3(wt.y, r512, phproj_, M, pe), ACT_NONE,
                         store_to(mem, rh, rawp("dit.phoneme_proj.bias"), 1.f, ph_mask), 1, pe, stt));
        HIPC(launch_to_split(mem, rh, wt.seqs.hi, wt.seqs.as(pk, satp(SITE_CROSS_KV)).lo, rh, M, kHidden, stt));
        EpiKV kv{k_text, v_text, kvtext_b_, B, kHeads, kDh, P};
        HIPC(gemm3_kv(ops3(wt.seqs, rh, kvtext_, M, pk), kv, pk, stt));
        HIPC(launch_headnorm(k_text, kBlocks, B, kHeads, P, kDh, 1e-6f, knc_, stt));
    }
    if (fork) HIPC(hipEventRecord(ev_join_, aux_));
    // ---- E1 style encoder (style.py:144-174) -------------------------------------------------
    if (R > 0) {
        const int M = B * R;
        HIPC(launch_len_mask(ref_len, ref_mask, B, R, st));
        HIPC(gemm_store(ops(ref, rowmap_plain(kLatent), style_in_, M), ACT_NONE,
                        store_to(w.x, r512, rawp("style_encoder.in_proj.bias"), style_scale_), 1, pc, st));
        if (run_encoder(st, style_, &w, B, R, ref_mask)) return 1;  // leaves RMSNorm(x; final_norm) in w.y
        float* seq = ref_seq_out ? ref_seq_out : w.seq;
        HIPC(gemm3_store(ops3(w.y, r512, style_out_, M, pe), ACT_NONE,
                         store_to(seq, rh, rawp("style_encoder.out_proj.bias"), 1.f, ref_mask), 1, pe, st));
        // ---- E3 cross KV for the reference tokens (dit.py:80-93) -------------------------------
        HIPC(launch_to_split(seq, rh, w.seqs.hi, w.seqs.as(pk, satp(SITE_CROSS_KV)).lo, rh, M, kHidden, st));
        EpiKV kv{k_ref, v_ref, kvref_b_, B, kHeads, kDh, R};
        HIPC(gemm3_kv(ops3(w.seqs, rh, kvref_, M, pk), kv, pk, st));
        HIPC(launch_headnorm(k_ref, kBlocks, B, kHeads, R, kDh, 1e-6f, knc_, st));
    }
    if (fork) HIPC(hipStreamWaitEvent(st, ev_join_, 0));  // join: everything after cond_encode sees both halves
    return 0;
}

// ---------------------------------------------------------------------------------------------
// D1/D3/D5/D8 modulation table: all AdaLN vectors for `rows` distinct timesteps in one pass
//   mod[r] = [ blk0: shift_msa scale_msa tanh(gate_msa) shift_mlp scale_mlp tanh(gate_mlp) | ... | final: scale shift ]
// (tiny-M GEMM chain on the fp32-A kernel; the tanh of the gates, dit.py:198,201, is applied once here
// instead of per output element in the residual epilogues)
// ---------------------------------------------------------------------------------------------
int Engine::modulation(hipStream_t st, const float* t_dev, int rows, float* sinb, float* t1, float* temb, float* e1,
                       float* semb, float* mod, float* ftab) {
    ProfTag ptag("mod");
    const int pc = prec_[SITE_COND];  // tiny-M chain on the fp32-A kernel: always split-bf16 unless the preset is plain bf16
    HIPC(launch_time_sinusoid(t_dev, sinb, rows, st));
    HIPC(gemm_store(ops(sinb, rowmap_plain(256), time0_, rows), ACT_SILU,
                    store_to(t1, rowmap_plain(kHidden), rawp("time_embedding.mlp.0.bias")), 1, pc, st));
    HIPC(gemm_store(ops(t1, rowmap_plain(kHidden), time2_, rows), ACT_NONE,
                    store_to(temb, rowmap_plain(kHidden), rawp("time_embedding.mlp.2.bias")), 1, pc, st));
    HIPC(gemm_store(ops(temb, rowmap_plain(kHidden), emb0_, rows), ACT_SILU,
                    store_to(e1, rowmap_plain(2 * kHidden), rawp("dit.emb_proj.0.bias")), 1, pc, st));
    // AdaLN consumes silu(emb) only (dit.py:20,36), so the SiLU is fused into this epilogue
    HIPC(gemm_store(ops(e1, rowmap_plain(2 * kHidden), emb2_, rows), ACT_SILU,
                    store_to(semb, rowmap_plain(kHidden), rawp("dit.emb_proj.2.bias")), 1, pc, st));
    HIPC(gemm_store(ops(semb, rowmap_plain(kHidden), modall_, rows), ACT_NONE,
                    store_to(mod, rowmap_plain(kModLd), modall_b_), 1, pc, st));
    HIPC(launch_tanh_gates(mod, rows, kModLd, kBlocks, kModPerBlock, kHidden, st));
    if (ftab) {   // LN-fold tables of every (step, block, site): W shift and W (1 + scale) on the weights the block GEMMs multiply
        const int pb = prec_[SITE_DIT_BLOCK];
        FoldSites fs{};
        fs.n = 0;
        fs.NF = kFoldNF;
        for (int l = 0; l < kBlocks; ++l) {
            const DitBlockW& b = blocks_[l];
            for (int site = 0; site < 2; ++site) {
                const PW& w = site ? b.ff13 : b.qkvgp;
                if (w.K != kHidden || w.N != (site ? 2 * kFF : 4 * kHeads * 128)) return fail("LN-fold: unexpected weight pack shape");
                FoldSite& f = fs.s[fs.n++];
                f.w = pb == PREC_F16 ? w.h16 : w.hi;
                f.wlo = pb == PREC_BF16X3 ? w.lo : nullptr;
                f.fmt = pb == PREC_F16 ? 0 : 1;
                f.N = w.N;
                f.shift_off = l * kModPerBlock + (site ? 3 : 0) * kHidden;
                f.scale_off = l * kModPerBlock + (site ? 4 : 1) * kHidden;
                f.out_off = l * kFoldPerBlock + (site ? 4L * kHeads * 128 : 0);
            }
        }
        HIPC(launch_fold_vectors(fs, mod, kModLd, rows, ftab, st));
    }
    return 0;
}

namespace {
struct ModWs {
    float *sinb, *t1, *temb, *e1, *semb, *mod, *ftab = nullptr;
    void plan(Bump& b, int rows, bool fold = false) {
        if (fold) ftab = b.take<float>((size_t)rows * 2 * kFoldNF);
        sinb = b.take<float>((size_t)rows * 256);
        t1 = b.take<float>((size_t)rows * kHidden);
        temb = b.take<float>((size_t)rows * kHidden);
        e1 = b.take<float>((size_t)rows * 2 * kHidden);
        semb = b.take<float>((size_t)rows * kHidden);
        mod = b.take<float>((size_t)rows * kModLd);
    }
};
struct CoreWs {
    float *h, *x, *qkvg, *part, *lnpart;
    float *rope_c, *rope_s;  // cos / sin of a caller-supplied angle table: per call (several calls may be in flight on different streams)
    SplitBuf gm1, gm2, y, o, ffh;
    SplitBuf qi, ki, vti, gi;   // attention operand images of the self part: [B][8][N][128] x 2, [B][8][128][pad8(N)], [M][960]
    size_t gm_elems, vt_elems;
    void plan(Bump& b, int B, int N) {
        const size_t M = (size_t)B * N;
        qi = take_split(b, M * kHeads * 128);
        ki = take_split(b, M * kHeads * 128);
        vt_elems = (size_t)B * kHeads * 128 * pad8(N);
        vti = take_split(b, vt_elems);
        gi = take_split(b, M * kHidden);
        gm_elems = (size_t)B * kConvG * (N + 2 * kConvPad) * kConvGs;
        h = b.take<float>(M * kHidden);
        x = b.take<float>(M * kHidden);
        qkvg = b.take<float>(M * 4 * kHidden);
        part = b.take<float>(M * kHidden * kSplitK);
        lnpart = b.take<float>(M * kLnGroups * 2);
        rope_c = b.take<float>((size_t)N * 64);
        rope_s = b.take<float>((size_t)N * 64);
        gm1 = take_split(b, gm_elems);
        gm2 = take_split(b, gm_elems);
        y = take_split(b, M * kHidden);
        o = take_split(b, M * kHidden);
        ffh = take_split(b, M * kFFp);
    }
};
}  // namespace

size_t Engine::denoise_core_bytes(int B, int N) const {
    Bump b(nullptr);
    CoreWs w;
    w.plan(b, B, N);
    return b.off + 256;
}

// ---------------------------------------------------------------------------------------------
// D0: one denoiser evaluation given a ready modulation table (model.py:97-100, dit.py:316-327)
// ---------------------------------------------------------------------------------------------
int Engine::denoise_core(hipStream_t st, const float* x_t, const uint8_t* mask, const float* mod, int mod_row0,
                         int mod_rstride, const float* k_ref, const float* v_ref, const uint8_t* ref_mask,
                         const float* k_text, const float* v_text, const uint8_t* ph_mask, const float* rope, int B,
                         int N, int R, int P, float* velocity, char* wsp, const CrossImg& ci, const float* ftab) {
    ProfTag ptag("dit");
    Bump bump(wsp);
    CoreWs w;
    w.plan(bump, B, N);
    const int M = B * N;
    const RowMap rh = rowmap_plain(kHidden);
    // operand formats: the block GEMMs run at SITE_DIT_BLOCK precision, the latent in-projection / conv pos-embed / velocity
    // head at SITE_COND; each activation buffer is written in the format of the GEMM that reads it
    const int pb = prec_[SITE_DIT_BLOCK], pc = prec_[SITE_COND], pcp = prec_[SITE_CONVPOS];
    unsigned* const sb = satp(SITE_DIT_BLOCK);
    const SplitBuf gm1 = w.gm1.as(pcp, satp(SITE_CONVPOS)), gm2 = w.gm2.as(pcp, satp(SITE_CONVPOS)), yb = w.y.as(pb, sb), ob = w.o.as(pb, sb), ffh = w.ffh.as(pb, sb);
    // D2 input embedding (dit.py:246-253): h = proj(x); x = mask*mish(conv2(mask*mish(conv1(mask*h)))) + h
    HIPC(gemm_store(ops(x_t, rowmap_plain(kLatent), inproj_, M), ACT_NONE,
                    store_to(w.h, rh, rawp("dit.input_embed.proj.bias")), 1, pc, st));
    HIPC(launch_convpos_pack(w.h, mask, gm1.hi, gm1.lo, B, N, kConvG, kConvCpg, kConvPad, kConvGs, st));
    // zero regions that no kernel of this function writes (pad frames of the conv image, pad columns of the FF hidden, pad key
    // columns of V^T): once per workspace use — the sampler's later steps find them as the first step left them
    const bool init_ws = !ws_ready_;
    ws_ready_ = ws_keep_;
    if (init_ws) {
        HIPC(hipMemsetAsync(w.gm2.hi, 0, w.gm_elems * 2, st));
        if (pcp == PREC_BF16X3) HIPC(hipMemsetAsync(w.gm2.lo, 0, w.gm_elems * 2, st));
    }
    {
        // grouped conv k=31 as B*G small GEMMs: z = b*G + g, rows = frames, K = 31 taps x 64 (padded) channels
        // one product per GROUP over the rows of the whole batch (z = g, row m = (b, frame)): B * N rows fill their 64-row tiles
        // (600 rows: 94 %) where one product per (utterance, group) left the second tile of N = 75 rows at 11 of 64 (SMTTS_CONVPOS_BY_GROUP=0)
        const long zs = (long)(N + 2 * kConvPad) * kConvGs;
        const RowMap am = convpos_by_group_ ? rowmap_batched(kConvGs, N, (long)kConvG * zs, 0) : rowmap_plain(kConvGs);
        const int rows = convpos_by_group_ ? B * N : N, nz = convpos_by_group_ ? kConvG : B * kConvG;
        Gemm3Operands g = ops3(w.gm1, am, conv1_, rows, pcp, 0, kConvCpg);
        g.a_z = zs;
        g.w_z = (long)kConvCpg * conv1_.K;
        g.w_zmod = kConvG;
        EpiConvPos<0> e1{nullptr, nullptr, rawp("dit.input_embed.conv_pos_embed.conv1.bias"), mask, kConvG, kConvCpg, N,
                         kConvPad, kConvGs, gm2.hi, gm2.lo, convpos_by_group_ ? 1 : 0};
        HIPC(gemm3_convpos(g, false, e1, nz, pcp, st));
        g = ops3(w.gm2, am, conv2_, rows, pcp, 0, kConvCpg);
        g.a_z = zs;
        g.w_z = (long)kConvCpg * conv2_.K;
        g.w_zmod = kConvG;
        EpiConvPos<0> e2{w.x, w.h, rawp("dit.input_embed.conv_pos_embed.conv2.bias"), mask, kConvG, kConvCpg, N,
                         kConvPad, kConvGs, nullptr, nullptr, convpos_by_group_ ? 1 : 0};
        HIPC(gemm3_convpos(g, true, e2, nz, pcp, st));
    }
    // zero the padded tail columns [2400, 2432) of the FF hidden once per call
    if (init_ws) {
        HIPC(hipMemsetAsync(w.ffh.hi, 0, (size_t)M * kFFp * 2, st));
        if (pb == PREC_BF16X3) HIPC(hipMemsetAsync(w.ffh.lo, 0, (size_t)M * kFFp * 2, st));
    }
    const float* rc = rope_dit_cos_;
    const float* rs = rope_dit_sin_;
    if (rope) {  // caller-supplied angle table (reference operator input, infer/onnx.py:42-47,122)
        HIPC(launch_rope_cossin(rope, w.rope_c, w.rope_s, N * 64, st));
        rc = w.rope_c;
        rs = w.rope_s;
    }
    // The AdaLN in front of each GEMM is fused into the kernel that produced the residual stream it normalises
    // (split-K reduction + gated residual + LayerNorm-modulate in one pass); only the very first one runs alone.
    if (join_pending_) {   // the modulation table is being computed on the side stream (sample)
        HIPC(hipStreamWaitEvent(st, ev_join_, 0));
        join_pending_ = false;
    }
    HIPC(launch_ln_modulate(w.x, nullptr, yb.hi, yb.lo, M, kHidden, 1e-6f, mod + 0 * kHidden, mod + 1 * kHidden, kModLd,
                            mod_row0, mod_rstride, N, st));
    // split-K exists to fill the chip at M = 600 (150 tiles of 64x64 for N = 960); the 3B-row CFG batches of the teacher
    // sampler (M = 1800: 435 tiles) fill it without, and the fused epilogue is cheaper than partials + reduce (495 -> 454 ms)
    // ... and so do several batches in flight (throughput tuning): there the unsplit GEMM + a separate AdaLN costs 2.3x fewer
    // workgroup-microseconds than three K slices + reduce, and that is what counts when other streams want the CUs
    static const bool splitk_tp = lab_env("SMTTS_SPLITK_TP") && atoi(lab_env("SMTTS_SPLITK_TP")) != 0;   // (A/B only)
    const bool unsplit = M > 1024 || (tuning_ == TUNE_THROUGHPUT && !splitk_tp);
    const int ks_out = unsplit ? 1 : ksplit_out_, ks_ff2 = unsplit ? 1 : ksplit_ff2_;
    if (!(attn_img_ && attn_epi_) && ensure_qkvg_unpadded()) return 1;
    // LN-fold (gemm.hpp LnFoldIn): inside the fused sampler — one modulation row for the whole batch, tables from modulation() —
    // the AdaLN between two block GEMMs lives in their epilogues; the first AdaLN of a step (above) and the final one (velocity
    // head, SITE_COND precision) keep their ln_modulate
    const bool fold = ln_fold_ && ftab && mod_rstride == 0 && attn_img_ && attn_epi_;   // (ftab: null under throughput tuning, see sample())
    const float* const mrow = mod + (long)mod_row0 * kModLd;            // this step's modulation row (fold path only)
    const float* const frow = fold ? ftab + (long)mod_row0 * 2 * kFoldNF : nullptr;   // [0]: W shift, [1]: W (1 + scale)
    auto fold_in = [&](int l, int site) {
        LnFoldIn f;
        f.part = w.lnpart; f.NP = kLnGroups; f.inv_c = 1.0f / kHidden; f.eps = 1e-6f;
        const long off = (long)l * kFoldPerBlock + (site ? 4L * kHeads * 128 : 0);
        f.wsh = frow + off;
        f.wc = frow + kFoldNF + off;
        return f;
    };
    for (int l = 0; l < kBlocks; ++l) {
        const DitBlockW& b = blocks_[l];
        const float* m = mod + (long)l * kModPerBlock;
        // D5 AdaLN-Zero (dit.py:19-25) already in w.y; D6 joint attention (dit.py:95-135)
        const bool epi = attn_img_ && attn_epi_;   // the GEMM's own epilogue writes the attention operands
        if (!epi)
            HIPC(gemm3_store(ops3(w.y, rh, b.qkvg, M, pb), ACT_NONE, store_to(w.qkvg, rowmap_plain(4 * kHidden), b.b_qkvg), 1,
                             pb, st));
        AttnArgs a{};
        a.q = w.qkvg; a.k = w.qkvg + kHidden; a.v = w.qkvg + 2 * kHidden; a.gate = w.qkvg + 3 * kHidden;
        a.bs = (long)N * 4 * kHidden; a.rs = 4 * kHidden;
        a.qw = b.qn; a.kw = b.kn; a.eps = 1e-6f;
        a.rope_cos = rc; a.rope_sin = rs; a.rot_dim = 64;
        const long lr = (long)l * B * kHeads * R * kDh, lp = (long)l * B * kHeads * P * kDh;
        a.k_ref = R > 0 ? k_ref + lr : nullptr; a.v_ref = R > 0 ? v_ref + lr : nullptr; a.R = R;
        a.k_text = P > 0 ? k_text + lp : nullptr; a.v_text = P > 0 ? v_text + lp : nullptr; a.P = P;
        a.mask_self = mask; a.mask_ref = ref_mask; a.mask_text = ph_mask;
        a.out = nullptr; a.out_hi = ob.hi; a.out_lo = ob.lo; a.obs = (long)N * kHidden; a.ors = kHidden;
        a.B = B; a.N = N; a.H = kHeads; a.dh = kDh;
        if (attn_img_) {
            const int pa = prec_[SITE_ATTN], Np = pad8(N);
            QkvPackArgs pk{};
            pk.qkvg = w.qkvg; pk.qw = b.qn; pk.kw = b.kn; pk.eps = 1e-6f; pk.q_scale = 1.0f / sqrtf((float)kDh);
            pk.rope_cos = rc; pk.rope_sin = rs; pk.rot_dim = 64; pk.prec = pa;
            pk.q = w.qi.hi; pk.q_lo = img_lo(pa, w.qi.lo); pk.k = w.ki.hi; pk.k_lo = img_lo(pa, w.ki.lo); pk.vt = w.vti.hi; pk.vt_lo = img_lo(pa, w.vti.lo);
            pk.g = w.gi.hi; pk.g_lo = img_lo(pa, w.gi.lo);
            pk.B = B; pk.N = N; pk.H = kHeads; pk.dh = kDh; pk.dhp = 128; pk.Np = Np;
            if (l == 0 && Np != N && init_ws) {   // pad key columns of V^T (the producer only writes n < N)
                HIPC(hipMemsetAsync(w.vti.hi, 0, w.vt_elems * 2, st));
                if (pa == PREC_BF16X3) HIPC(hipMemsetAsync(w.vti.lo, 0, w.vt_elems * 2, st));
            }
            if (epi) {
                EpiQKV eq{b.b_qkvgp, pk.qw, pk.kw, pk.rope_cos, pk.rope_sin, pk.eps, pk.q_scale, pk.rot_dim, pa,
                          pk.q, pk.q_lo, pk.k, pk.k_lo, pk.vt, pk.vt_lo, pk.g, pk.g_lo, N, kHeads, kDh, 128, Np};
                if (fold && l > 0) eq.fold = fold_in(l, 0);   // w.y = x (1 + scale_msa), written by the previous block's FF2 epilogue
                HIPC(gemm3_qkv(ops3(w.y, rh, b.qkvgp, M, pb), eq, pb, st));
            } else {
                HIPC(launch_qkv_pack(pk, st));
            }
            AttnImg ai{};
            ai.prec = pa;
            ai.q = pk.q; ai.q_lo = w.qi.lo; ai.k = pk.k; ai.k_lo = w.ki.lo; ai.vt = pk.vt; ai.vt_lo = w.vti.lo;   // (the consumer reads the lo ARRAYS, split format only)
            ai.g = pk.g; ai.g_lo = sm_lo_for(pa, w.gi.lo);
            if (ci.Cp > 0) {
                const long lc = (long)l * B * kHeads * ci.Cp * 128;
                ai.kc = ci.kc + lc; ai.vtc = ci.vtc + lc;
                if (pa == PREC_BF16X3) { ai.kc_lo = ci.kc_lo + lc; ai.vtc_lo = ci.vtc_lo + lc; }
            }
            ai.mask_self = mask; ai.mask_ref = ref_mask; ai.mask_text = ph_mask;
            ai.out_hi = ob.hi; ai.out_lo = ob.lo; ai.ors = kHidden;
            ai.B = B; ai.N = N; ai.H = kHeads; ai.dh = kDh; ai.Np = Np; ai.R = R; ai.P = P; ai.Rp = ci.Rp; ai.Cp = ci.Cp;
            HIPC(launch_attention_img(ai, st));
        } else {
#ifdef SMTTS_TEST_KERNELS   // the fp32 VALU reference attention (attention.hip): test builds only
            a.prenormed = 1;
            HIPC(launch_qk_prep(a, st));
            HIPC(launch_attention(a, st));
#else
            return fail("the fp32 VALU reference attention is not part of this build (make TEST_KERNELS=1)");
#endif
        }
        // to_out + mask + gated residual (dit.py:117-118,198), then the MLP AdaLN (dit.py:199)
        EpiResid<0> r1{w.x, rh, nullptr, m + 2 * kHidden, kModLd, mod_row0, mod_rstride, N, mask};
        NextLN ln1{m + 3 * kHidden, m + 4 * kHidden, yb.hi, yb.lo};
        const float* const mr = mrow + (long)l * kModPerBlock;
        if (fold) {
            EpiResidLN e1{w.x, rh, nullptr, mr + 2 * kHidden, mask, mr + 4 * kHidden, yb.hi, yb.lo, kHidden, w.lnpart, kLnGroups};
            HIPC(gemm3_resid_ln(ops3(w.o, rh, b.out, M, pb), e1, pb, st));
        } else if (ks_out > 1) {
            HIPC(gemm3_resid_splitk(ops3(w.o, rh, b.out, M, pb), r1, w.part, ks_out, pb, st, ln1));
        } else {
            HIPC(gemm3_resid(ops3(w.o, rh, b.out, M, pb), 1, r1, pb, st));
            HIPC(launch_ln_modulate(w.x, nullptr, ln1.yhi, ln1.ylo, M, kHidden, 1e-6f, ln1.shift, ln1.scale, kModLd, mod_row0,
                                    mod_rstride, N, st));
        }
        // D7 feed-forward (dit.py:199-201)
        EpiSwiGLU sw{nullptr, kFFp, b.b1, b.b3, ffh.hi, ffh.lo};
        if (fold) sw.fold = fold_in(l, 1);
        HIPC(gemm3_swiglu(ops3(w.y, rh, b.ff13, M, pb), sw, pb, st));
        // w2 + gated residual, then the next block's attention AdaLN — or D8's final AdaLN (chunk order scale, shift:
        // dit.py:37) after the last block
        EpiResid<0> r2{w.x, rh, b.b2, m + 5 * kHidden, kModLd, mod_row0, mod_rstride, N, nullptr};
        const float* mn = m + kModPerBlock;  // next block's modulation (or the final norm's [scale | shift])
        const SplitBuf yv = w.y.as(pc, satp(SITE_COND));  // the final AdaLN feeds the velocity head (SITE_COND)
        NextLN ln2 = l + 1 < kBlocks ? NextLN{mn + 0 * kHidden, mn + 1 * kHidden, yb.hi, yb.lo}
                                     : NextLN{mn + kHidden, mn, yv.hi, yv.lo};
        if (fold && l + 1 < kBlocks) {
            EpiResidLN e2{w.x, rh, b.b2, mr + 5 * kHidden, nullptr, mr + kModPerBlock + kHidden, yb.hi, yb.lo, kHidden, w.lnpart, kLnGroups};
            HIPC(gemm3_resid_ln(ops3(w.ffh, rowmap_plain(kFFp), b.ff2, M, pb), e2, pb, st));
        } else if (ks_ff2 > 1) {
            HIPC(gemm3_resid_splitk(ops3(w.ffh, rowmap_plain(kFFp), b.ff2, M, pb), r2, w.part, ks_ff2, pb, st, ln2));
        } else {
            HIPC(gemm3_resid(ops3(w.ffh, rowmap_plain(kFFp), b.ff2, M, pb), 1, r2, pb, st));
            HIPC(launch_ln_modulate(w.x, nullptr, ln2.yhi, ln2.ylo, M, kHidden, 1e-6f, ln2.shift, ln2.scale, kModLd, mod_row0,
                                    mod_rstride, N, st));
        }
    }
    // velocity head (model.py:100) on the final AdaLN output
    HIPC(gemm3_store(ops3(w.y, rh, velocity_, M, pc), ACT_NONE, store_to(velocity, rowmap_plain(kLatent), rawp("velocity.bias")),
                     1, pc, st));
    return 0;
}

// The unpadded QKVG pack of the DiT blocks, read by the A/B attention paths only: packed when one of them first runs (a test hook
// can flip the path after finalize), from the fp32 tensors that stay resident.  Synchronises the device: never on the product path.
int Engine::ensure_qkvg_unpadded() {
    if (blocks_.empty() || qkvg_unpadded_ready_) return 0;
    HIPC(hipDeviceSynchronize());
    struct PackMode { bool& f; bool old; explicit PackMode(bool& b) : f(b), old(b) { f = true; } ~PackMode() { f = old; } } pm(packing_);
    for (DitBlockW& b : blocks_) {
        if (b.qkvg.N) continue;   // (packed by finalize_dit when an A/B path was selected before it)
        b.qkvg = pack_rows({b.name + ".attn.to_q.weight", b.name + ".attn.to_k_self.weight", b.name + ".attn.to_v_self.weight",
                            b.name + ".attn.gate.weight"});
        if (!b.qkvg.N) {
            // a block failed (out of memory): nothing half-built may look finished to the next call — the packs made so far stay in
            // pack_allocs_ (freed by the next finalize) but are unlinked, and the flag stays down
            for (DitBlockW& c : blocks_) c.qkvg = PW();
            return fail("DiT block " + b.name + ": unpadded QKVG pack failed (" + err_ + ")");
        }
    }
    HIPC(hipDeviceSynchronize());
    qkvg_unpadded_ready_ = true;
    return 0;
}

// Cross-KV cache of all 12 layers in the attention kernel's operand format: built once per sampler call (or per denoise_step
// call) from the fp32 rank-5 tensors the C ABI hands over — 25 MB of traffic against 4 x 12 attention launches that then DMA it.
size_t Engine::cross_img_bytes(int B, int R, int P) const {
    const size_t Cp = (size_t)pad8(R) + pad8(P);
    return 4 * ((size_t)kBlocks * B * kHeads * Cp * 128 * 2 + 256) + 256;   // Kc, Vc^T, each hi + lo
}
int Engine::pack_cross(hipStream_t st, const float* k_ref, const float* v_ref, const float* k_text, const float* v_text, int B,
                       int R, int P, char* ws, CrossImg& ci) {
    ci = CrossImg();
    if (!attn_img_ || (R <= 0 && P <= 0)) return 0;
    const int pa = prec_[SITE_ATTN];
    Bump bump(ws);
    ci.Rp = pad8(R > 0 ? R : 0);
    ci.Cp = ci.Rp + pad8(P > 0 ? P : 0);
    const size_t n = (size_t)kBlocks * B * kHeads * ci.Cp * 128;
    ci.kc = bump.take<bf16_t>(n); ci.kc_lo = bump.take<bf16_t>(n);
    ci.vtc = bump.take<bf16_t>(n); ci.vtc_lo = bump.take<bf16_t>(n);
    CrossPackArgs p{};
    p.k_ref = k_ref; p.v_ref = v_ref; p.k_text = k_text; p.v_text = v_text;
    p.kc = ci.kc; p.kc_lo = img_lo(pa, ci.kc_lo); p.vtc = ci.vtc; p.vtc_lo = img_lo(pa, ci.vtc_lo);
    p.prec = pa; p.L = kBlocks; p.B = B; p.H = kHeads; p.dh = kDh; p.dhp = 128;
    p.R = R > 0 ? R : 0; p.P = P > 0 ? P : 0; p.Rp = ci.Rp; p.Cp = ci.Cp;
    ProfTag ptag("dit");
    HIPC(launch_cross_pack(p, st));
    return 0;
}

size_t Engine::denoise_ws_bytes(int B, int N, int R, int P, int rows) const {
    Bump b(nullptr);
    ModWs m;
    m.plan(b, rows);
    return b.off + 256 + denoise_core_bytes(B, N) + cross_img_bytes(B, R, P);
}

int Engine::denoise_step(hipStream_t st, const float* x_t, const uint8_t* mask, const float* t, const float* k_ref,
                         const float* v_ref, const uint8_t* ref_mask, const float* k_text, const float* v_text,
                         const uint8_t* ph_mask, const float* rope, int B, int N, int R, int P, float* velocity,
                         void* ws, size_t ws_bytes) {
    DeepScope deep_scope(gemm_deep_);
    if (!dit_ready_) return fail("denoise_step: DiT weights not finalized");
    if (N > kMaxPos) return fail("denoise_step: sequence longer than the rope table (4096)");
    if (ws_bytes < denoise_ws_bytes(B, N, R, P, B)) return fail("denoise_step: workspace too small");
    HIPC(hipSetDevice(device_));
    Bump bump(ws);
    ModWs m;
    m.plan(bump, B);
    if (modulation(st, t, B, m.sinb, m.t1, m.temb, m.e1, m.semb, m.mod)) return 1;
    char* core = static_cast<char*>(ws) + ((bump.off + 255) & ~size_t(255));
    CrossImg ci;
    if (pack_cross(st, k_ref, v_ref, k_text, v_text, B, R, P, core + ((denoise_core_bytes(B, N) + 255) & ~size_t(255)), ci)) return 1;
    return denoise_core(st, x_t, mask, m.mod, 0, 1, k_ref, v_ref, ref_mask, k_text, v_text, ph_mask, rope, B, N, R, P,
                        velocity, core, ci);
}

// ---------------------------------------------------------------------------------------------
// S1 / S2 samplers
// ---------------------------------------------------------------------------------------------
void alpha_sigma_host(float t, float& a, float& s) {
    // reference infer/onnx.py:31-39 : float64 math on the float32 t, cast to float32 at the end
    double td = (double)t;
    const double eps = 1e-5;
    td = td < eps ? eps : (td > 1.0 - eps ? 1.0 - eps : td);
    double c = std::cos(M_PI / 2.0 * td);
    double a2 = c * c;
    double lsnr = std::log(a2 / (1.0 - a2)) + 2.0 * std::log(0.5);
    double asq = 1.0 / (1.0 + std::exp(-lsnr));
    a = (float)std::sqrt(asq);
    s = (float)std::sqrt(1.0 - asq);
}
static float linspace10(int i, int n) {
    // np.linspace(1, 0, n, dtype=float32): float64 arithmetic, endpoint exact, cast to float32
    if (n <= 1) return 1.0f;
    if (i == n - 1) return 0.0f;
    double step = -1.0 / (double)(n - 1);
    return (float)(1.0 + step * (double)i);
}

namespace {
struct SampleWs {
    float *ts, *xt, *v, *x, *nz, *xt3, *v3;
    void plan(Bump& b, int B, int N, int n_steps, int cfg) {
        const size_t e = (size_t)B * N * kLatent;
        ts = b.take<float>(n_steps);
        xt = b.take<float>(e);
        v = b.take<float>(e);
        x = b.take<float>(e);
        nz = b.take<float>(e);
        xt3 = b.take<float>(cfg ? 3 * e : 1);
        v3 = b.take<float>(cfg ? 3 * e : 1);
    }
};
}  // namespace

size_t Engine::sample_ws_bytes(int B, int N, int R, int P, int n_steps, int cfg) const {
    Bump b(nullptr);
    SampleWs s;
    s.plan(b, B, N, n_steps, cfg);
    ModWs m;
    m.plan(b, n_steps, ln_fold_now() && (long)(cfg ? 3 * B : B) * N <= kFoldMaxRows);
    return b.off + 256 + denoise_core_bytes(cfg ? 3 * B : B, N) + cross_img_bytes(cfg ? 3 * B : B, R, P);
}

int Engine::sample(hipStream_t st, int mode, int n_steps, int cfg, float s_text, float s_spk, const uint8_t* mask,
                   const float* k_ref, const float* v_ref, const uint8_t* ref_mask, const float* k_text,
                   const float* v_text, const uint8_t* ph_mask, int B, int N, int R, int P, const float* noise,
                   uint64_t seed, float* x_out, float* steps_out, void* ws, size_t ws_bytes) {
    DeepScope deep_scope(gemm_deep_);
    if (!dit_ready_) return fail("sample: DiT weights not finalized");
    if (n_steps < 1) return fail("sample: n_steps must be >= 1");
    if (N > kMaxPos) return fail("sample: sequence longer than the rope table (4096)");
    if (ws_bytes < sample_ws_bytes(B, N, R, P, n_steps, cfg)) return fail("sample: workspace too small");
    HIPC(hipSetDevice(device_));
    Bump bump(ws);
    SampleWs s;
    s.plan(bump, B, N, n_steps, cfg);
    ModWs m;
    // (the fold replaces split-K partials + reduce; the 3B-row CFG batches of the teacher run unsplit + ln_modulate already, and there
    // the table kernel — every block weight once per four steps — and the heavier epilogue cost more than the norm launches: 239.0 vs
    // 240.6 ms per 128-step batch, profiles/r06l_ab_teacher_fold.txt)
    m.plan(bump, n_steps, ln_fold_now() && (long)(cfg ? 3 * B : B) * N <= kFoldMaxRows);
    char* core = static_cast<char*>(ws) + ((bump.off + 255) & ~size_t(255));
    const long e = (long)B * N * kLatent;

    std::vector<float> ts(n_steps), al(n_steps), sg(n_steps);
    for (int i = 0; i < n_steps; ++i) {
        ts[i] = linspace10(i, n_steps);
        alpha_sigma_host(ts[i], al[i], sg[i]);
    }
    // the same float32(np.linspace(1, 0, n)) values, generated on the device: a host -> device copy of the stack array would
    // need a stream synchronisation here, which drains the queue and leaves the GPU waiting on kernel launches for the
    // first DiT block of every batch (~0.1-0.5 ms of idle time in the rocprof timeline)
    HIPC(launch_linspace10(s.ts, n_steps, st));
    // t is shared by the whole batch -> every AdaLN vector of every step in one pass (SURVEY §7 hard part 3)
    // The table does not depend on the batch's data, only the first AdaLN needs it: in latency tuning it is computed on the
    // engine's side stream while the main stream packs the cross-KV images and embeds the first step's input (0.2 ms of tiny-M
    // GEMMs that fill a fraction of the chip); the main stream joins right before its first ln_modulate (denoise_core).
    // (an early return between the fork and the first denoise_core must not leave the flag set for the next call, nor the side
    // stream's writes into this workspace unordered against whatever the caller does next on `st`: ADVICE r3)
    struct JoinGuard {
        Engine* e; hipStream_t st;
        ~JoinGuard() { if (e->join_pending_) { (void)hipStreamWaitEvent(st, e->ev_join_, 0); e->join_pending_ = false; } }
    } join_guard{this, st};
    if (dual_stream_ && !prof_on_) {
        if (ensure_aux(st)) return 1;
        HIPC(hipEventRecord(ev_fork_, st));
        HIPC(hipStreamWaitEvent(aux_, ev_fork_, 0));
        const int mod_rc = modulation(aux_, s.ts, n_steps, m.sinb, m.t1, m.temb, m.e1, m.semb, m.mod, m.ftab);
        HIPC(hipEventRecord(ev_join_, aux_));   // recorded even when the chain failed half-way: the guard joins what was enqueued
        join_pending_ = true;
        if (mod_rc) return 1;
    } else if (modulation(st, s.ts, n_steps, m.sinb, m.t1, m.temb, m.e1, m.semb, m.mod, m.ftab)) {
        return 1;
    }

    const int Bd = cfg ? 3 * B : B;
    struct KeepWs { Engine* e; explicit KeepWs(Engine* e_) : e(e_) { e->ws_ready_ = false; e->ws_keep_ = true; } ~KeepWs() { e->ws_ready_ = e->ws_keep_ = false; } } keep_ws(this);
    CrossImg ci;   // the cross-KV cache in the attention kernel's operand format: once per call, read by every step
    if (pack_cross(st, k_ref, v_ref, k_text, v_text, Bd, R, P, core + ((denoise_core_bytes(Bd, N) + 255) & ~size_t(255)), ci)) return 1;
    // mask for 3B rows when cfg: caller passes mask with B rows; replicate by pointer arithmetic is impossible,
    // so the cfg path expects `mask` to already hold 3B rows (documented in the header).
    auto eval_velocity = [&](int step) -> int {
        if (!cfg)
            return denoise_core(st, s.xt, mask, m.mod, step, 0, k_ref, v_ref, ref_mask, k_text, v_text, ph_mask, nullptr,
                                B, N, R, P, s.v, core, ci, m.ftab);
        for (int r = 0; r < 3; ++r)
            HIPC(hipMemcpyAsync(s.xt3 + r * e, s.xt, e * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (denoise_core(st, s.xt3, mask, m.mod, step, 0, k_ref, v_ref, ref_mask, k_text, v_text, ph_mask, nullptr, Bd,
                         N, R, P, s.v3, core, ci, m.ftab))
            return 1;
        HIPC(launch_cfg_combine(s.v3, s.v, s_text, s_spk, e, st));
        return 0;
    };

    if (mode == 0) {
        HIPC(hipMemsetAsync(s.x, 0, e * sizeof(float), st));
        for (int i = 0; i < n_steps; ++i) {
            const float* nz = noise ? noise + (long)i * e : s.nz;
            if (!noise) HIPC(launch_randn(s.nz, e, seed, (uint64_t)i, st));
            HIPC(launch_axpby(s.xt, s.x, nz, al[i], sg[i], e, st));       // x_t = a x + s eps
            if (eval_velocity(i)) return 1;
            HIPC(launch_axpby(s.x, s.xt, s.v, al[i], -sg[i], e, st));     // x = a x_t - s v
            if (steps_out) HIPC(hipMemcpyAsync(steps_out + (long)i * e, s.x, e * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    } else {
        const float* nz = noise ? noise : s.nz;
        if (!noise) HIPC(launch_randn(s.nz, e, seed, 0, st));
        HIPC(hipMemsetAsync(s.x, 0, e * sizeof(float), st));
        HIPC(launch_axpby(s.xt, s.x, nz, 0.f, sg[0], e, st));             // x_1 = sigma(1) eps
        for (int i = 0; i < n_steps; ++i) {
            if (eval_velocity(i)) return 1;
            const bool last = i + 1 == n_steps;
            HIPC(launch_ode_step(s.xt, s.v, s.x, al[i], sg[i], last ? al[i] : al[i + 1], last ? sg[i] : sg[i + 1], e, st));
            if (steps_out) HIPC(hipMemcpyAsync(steps_out + (long)i * e, s.x, e * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    HIPC(hipMemcpyAsync(x_out, s.x, e * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Codec (V1 decode / V0 encode).  Channels-last images [B][kCodecPad + T][C]; the zero pad frames in
// front of each batch item implement the causal left padding of every conv, so strided / transposed /
// k-tap convs all become plain GEMMs over overlapping rows.
// ---------------------------------------------------------------------------------------------
// Stage chain (codec_ffn_wave.hip): every block of a C = 32 stage on each 32-frame tile back to back — the stage's image is read once
// and written once instead of once per block.  Returns 1 when the stage was enqueued (images swapped); 0 = not eligible, the
// caller runs the blocks one by one (bit-identical results either way: tests/test_codec_gpu.py::test_alternative_paths_*).
int Engine::codec_stage_chain(hipStream_t st, const CodecStageW& sg, float** xp, float** xaltp, int B, int T, int C) {
    const int F = cspec_.ffn_mult * C, nb = (int)sg.blocks.size();
    if (!stage_chain_ || !fused_ffn_ || !block_wave_ || nb < chain_min_blocks_ || nb > 3) return 0;
    int pf = prec_[SITE_CODEC_FFN];
    for (const CodecBlockW& b : sg.blocks) {
        if (pf == PREC_F16 && !b.f16_ok) return 0;   // an uncertified block runs split-bf16: per-block path
        if (b.w1.K != C || b.w2.K != F) return 0;
    }
    if (!codec_chain_wave_ok(C, F, cspec_.kernel, T, pf, nb)) return 0;
    {
        // Every wave of the chain walks ONE contiguous run of tiles plus a warm-up tile that is computed and not stored: with short runs
        // (small decodes: one 10 s utterance is 3 tiles per wave) the warm-up is a third of the work.  Measured (ADVICE r5,
        // profiles/r06q_chain_small_batch.txt, B x 10 s, chain vs one launch per block): B = 1 1.567 vs 1.554 ms, B = 4 (10 tiles per
        // wave) 3.538 vs 3.517, B = 8 (20) 5.923 vs 5.994 -> the chain from 16 tiles per wave up.
        const long waves = (long)(persist_cus_ > 0 ? persist_cus_ : num_cus_) * 12;
        if (chain_min_run_ > 0 && (long)B * T / 32 < chain_min_run_ * waves) return 0;
    }
    CodecChainBlock cb[3];
    for (int i = 0; i < nb; ++i) {
        const CodecBlockW& b = sg.blocks[i];
        cb[i] = CodecChainBlock{b.norm_w, b.dw_w, b.dw_b, b.gamma, b.ffn_norm_w, pf == PREC_F16 ? b.w1.h16 : b.w1.hi, b.w1.K, b.b1,
                                pf == PREC_F16 ? b.w2.h16 : b.w2.hi, b.b2, b.ffn_gamma};
    }
    const RowMap img = rowmap_batched(C, T, (long)(kCodecPad + T) * C, (long)kCodecPad * C);
    const hipError_t e = launch_codec_chain_wave(*xp, *xaltp, img, cb, nb, B * T, C, F, cspec_.kernel, cspec_.eps, pf, st);
    if (e != hipSuccess) { fail_hip(e, "codec stage chain"); return -1; }
    float* t = *xp; *xp = *xaltp; *xaltp = t;
    return 1;
}

int Engine::codec_block(hipStream_t st, const CodecBlockW& w, float** xp, float** xaltp, float* nbuf, bf16_t* n2hi,
                        bf16_t* n2lo, bf16_t* hhi, bf16_t* hlo, int B, int T, int C, size_t n2_elems) {
    const int M = B * T, pad = kCodecPad;
    float* x = *xp;
    const RowMap img = rowmap_batched(C, T, (long)(pad + T) * C, (long)pad * C);
    const int F = cspec_.ffn_mult * C;
    const RowMap rc = rowmap_plain(C), rf = rowmap_plain(F);
    // a block whose fused-kernel range could not be certified never runs those kernels at fp16 (they do not clamp), whoever drives the C ABI
    const int pf = (prec_[SITE_CODEC_FFN] == PREC_F16 && !w.f16_ok) ? PREC_BF16X3 : prec_[SITE_CODEC_FFN];
    unsigned* const sf = satp(SITE_CODEC_FFN);
    bf16_t* const n2lo_f = sm_lo_for(pf, n2lo, sf);  // (hi, lo) pairs handed to producers: the format the FFN GEMMs read
    auto wsel = [pf](const PW& w) { return pf == PREC_F16 ? w.h16 : w.hi; };
    bool n2_done = false;  // the FFN's normalised input was already produced by the fused depthwise-conv kernel
    // narrowest stages: the whole block in one pass over the image (mixer + FFN, codec_ffn_wave.hip MIX kernels)
    if (fused_ffn_ && block_wave_ && codec_block_wave_ok(C, F, cspec_.kernel, T, pf) && w.w1.K == C && w.w2.K == F) {
        HIPC(launch_codec_block_wave(x, *xaltp, img, w.norm_w, w.dw_w, w.dw_b, w.gamma, w.ffn_norm_w, wsel(w.w1), w.w1.lo, w.w1.K, w.b1,
                                     wsel(w.w2), w.w2.lo, w.b2, w.ffn_gamma, M, C, F, cspec_.kernel, cspec_.eps, pf, st));
        *xp = *xaltp;
        *xaltp = x;
        return 0;
    }
    // mixer: RMSNorm -> causal depthwise conv -> LayerScale residual
    if (C <= 256 && 256 % (C / 4) == 0 && cspec_.kernel <= 7 && fused_ffn_) {  // narrow stages: one out-of-place kernel, then swap images
        HIPC(launch_mixer_fused(x, *xaltp, w.norm_w, w.dw_w, w.dw_b, w.gamma, B, T, C, cspec_.kernel, pad, cspec_.eps, st));
        *xp = *xaltp;
        *xaltp = x;
        x = *xp;
    } else if (fused_ffn_ && mixer_wide_ && mixer_wide_ok(C, cspec_.kernel) && F % 64 == 0 && (size_t)M * C <= n2_elems) {
        // wide stages: mixer + the FFN's RMSNorm in one out-of-place pass (no normalised fp32 image in between), then swap images
        HIPC(launch_mixer_wide(x, *xaltp, w.norm_w, w.dw_w, w.dw_b, w.gamma, w.ffn_norm_w, n2hi, n2lo_f, B, T, C, cspec_.kernel, pad,
                               cspec_.eps, st));
        *xp = *xaltp;
        *xaltp = x;
        x = *xp;
        n2_done = true;
    } else {
        HIPC(launch_rmsnorm(x, img, nbuf, nullptr, nullptr, img, M, C, cspec_.eps, w.norm_w, st));
        if (fused_ffn_ && C % 64 == 0 && F % 64 == 0 && C <= 2048) {  // + the FFN's RMSNorm of the updated rows (split pair n2)
            HIPC(launch_dwconv_resid_rms(x, nbuf, w.dw_w, w.dw_b, w.gamma, B, T, C, cspec_.kernel, pad, cspec_.eps, w.ffn_norm_w,
                                         n2hi, n2lo_f, rowmap_plain(C), st));
            n2_done = true;
        } else {
            HIPC(launch_dwconv_resid(x, nbuf, w.dw_w, w.dw_b, w.gamma, B, T, C, cspec_.kernel, pad, st));
        }
    }
    // FFN: RMSNorm -> Linear 4x -> GELU -> Linear -> LayerScale residual
    if (fused_ffn_ && (C == 32 || C == 64) && F == 4 * C && w.w1.K == C && w.w2.K == F) {
        // narrowest stages: all weights LDS-resident, one wave per 32 frames, hidden stays in registers (codec_ffn_wave.hip)
        HIPC(launch_codec_ffn_wave(x, img, w.ffn_norm_w, wsel(w.w1), w.w1.lo, w.w1.K, w.b1, wsel(w.w2), w.w2.lo, w.b2, w.ffn_gamma, M,
                                   C, F, cspec_.eps, pf, st));
        return 0;
    }
    if (w.w2t.N && fused_ffn_) {
        // C = 128 / 256: weights stream through an LDS ring, hidden in registers (codec_ffn_stream.hip)
        HIPC(launch_codec_ffn_stream(x, img, w.ffn_norm_w, wsel(w.w1), w.w1.lo, w.b1, wsel(w.w2t), w.w2t.lo, w.b2, w.ffn_gamma, M, C, F,
                                     cspec_.eps, pf, st));
        return 0;
    }
    // wide stages: two gemm3 launches; n2 / hidden are split bf16 pairs
    SplitBuf hid{hhi, hlo};
    if (F % 64 != 0) {  // hidden width not a whole k-tile (only tiny test specs): all-fp32-A path, hidden kept fp32
        float* hf = reinterpret_cast<float*>(hhi);  // hhi/hlo are adjacent: 2 x F x M bf16 = F x M fp32
        if (reinterpret_cast<char*>(hlo) < reinterpret_cast<char*>(hhi) + (size_t)M * F * 2) return fail("codec ws layout");
        HIPC(launch_rmsnorm(x, img, nbuf, nullptr, nullptr, img, M, C, cspec_.eps, w.ffn_norm_w, st));
        HIPC(gemm_store(ops(nbuf, img, w.w1, M), ACT_GELU, store_to(hf, rf, w.b1), 1, pf, st));
        EpiResid<0> r0{x, img, w.b2, w.ffn_gamma, 0, 0, 0, 1, nullptr};
        HIPC(gemm_resid(ops(hf, rf, w.w2, M), 2, r0, pf, st));
        return 0;
    }
    if (C % 64 == 0) {
        SplitBuf n2{n2hi, n2lo};
        if (!n2_done) HIPC(launch_rmsnorm(x, img, nullptr, n2hi, n2lo_f, rc, M, C, cspec_.eps, w.ffn_norm_w, st));
        HIPC(gemm3_store(ops3(n2, rc, w.w1, M, pf), ACT_GELU, store_split_to(hid.as(pf, sf), rf, w.b1), 1, pf, st));
    } else {  // K = C < 64 (last stage, C = 32): one k-tile on the fp32-A kernel, still writing the split hidden
        HIPC(launch_rmsnorm(x, img, nbuf, nullptr, nullptr, img, M, C, cspec_.eps, w.ffn_norm_w, st));
        HIPC(gemm_store(ops(nbuf, img, w.w1, M), ACT_GELU, store_split_to(hid.as(pf, sf), rf, w.b1), 1, pf, st));
    }
    EpiResid<0> r{x, img, w.b2, w.ffn_gamma, 0, 0, 0, 1, nullptr};
    {
        // M = 600 against a wide-K second product (the coarsest stage: 600 x 2048 x 8192): 160x128 tiles cut K so that ONE round
        // of workgroups fills the chip; fp32 partials go to the (now dead) n2 buffer, one pass reduces them in a fixed order
        // and applies bias / LayerScale / residual (same deterministic scheme as the DiT projections)
        extern int g_gemm3_t160;
        const long t160 = (long)((M + 159) / 160) * ((C + 127) / 128);
        const int nk = F / 64;
        int splits = t160 > 0 && t160 <= 128 ? (int)(256 / t160) : 1;
        if (splits > nk / 8) splits = nk / 8;
        if (g_gemm3_t160 && M > 480 && M <= 640 && splits >= 2 && (size_t)splits * M * C * 2 <= n2_elems && C % 4 == 0) {
            float* part = reinterpret_cast<float*>(n2hi);  // n2hi holds n2_elems bf16 = n2_elems / 2 floats
            HIPC(gemm3_resid_splitk(ops3(hid, rf, w.w2, M, pf), r, part, splits, pf, st, NextLN(), G3_160x128));
            return 0;
        }
    }
    {
        // fewer rows still (the codec ENCODER's coarse stages on 2-s references: 120 x 2048 x 8192 is 64 tiles that each walk the
        // whole K: 47 us at 0.7 TB/s of weights): K slices on 64x64 tiles + the same reduce
        const int splits = M <= 480 ? small_m_splits(M, F, 8, 1) : 1;
        if (splits >= 2 && fused_ffn_ && (size_t)splits * M * C * 2 <= n2_elems && C % 4 == 0) {
            float* part = reinterpret_cast<float*>(n2hi);
            HIPC(gemm3_resid_splitk(ops3(hid, rf, w.w2, M, pf), r, part, splits, pf, st, NextLN(), G3_64x64));
            return 0;
        }
    }
    HIPC(gemm3_resid(ops3(hid, rf, w.w2, M, pf), 2, r, pf, st));
    return 0;
}

namespace {
struct CodecWs {
    float *xa, *xb, *nb, *lat;
    bf16_t *n2hi, *n2lo, *hhi, *hlo;
    void plan(Bump& b, size_t max_img, size_t max_rows_c, size_t max_hidden, size_t lat_elems) {
        xa = b.take<float>(max_img);
        xb = b.take<float>(max_img);
        nb = b.take<float>(max_img);
        n2hi = b.take<bf16_t>(max_rows_c);
        n2lo = b.take<bf16_t>(max_rows_c);
        hhi = b.take<bf16_t>(max_hidden);
        hlo = b.take<bf16_t>(max_hidden);
        lat = b.take<float>(lat_elems);
    }
};
}  // namespace

size_t Engine::decode_ws_bytes(int B, int T) const {
    const CodecSpecC& s = cspec_;
    const int S = s.n_ratios + 1;
    size_t max_img = 0, max_hid = 0;
    long Ti = T;
    for (int i = 0; i < S; ++i) {
        if (i > 0) Ti *= s.ratios[i - 1];
        size_t C = (size_t)s.n_filters << (S - 1 - i);
        size_t img = (size_t)B * (kCodecPad + Ti) * C;
        max_img = img > max_img ? img : max_img;
        size_t hid = (size_t)B * Ti * C * s.ffn_mult;
        max_hid = hid > max_hid ? hid : max_hid;
    }
    Bump b(nullptr);
    CodecWs w;
    w.plan(b, max_img, max_img, max_hid, (size_t)B * (kCodecPad + T) * s.latent_dim);
    return b.off + 256;
}

int Engine::codec_decode(hipStream_t st, const float* latents, int B, int T, float* audio, void* ws, size_t ws_bytes) {
    DeepScope deep_scope(gemm_deep_);
    PersistScope persist_scope(persist_cus_);
    if (!dec_.ready) return fail("codec_decode: decoder weights not finalized");
    if (ws_bytes < decode_ws_bytes(B, T)) return fail("codec_decode: workspace too small");
    HIPC(hipSetDevice(device_));
    const CodecSpecC& s = cspec_;
    const int S = s.n_ratios + 1, pad = kCodecPad, Kc = s.kernel, L = s.latent_dim;
    const int pcv = prec_[SITE_CODEC_CONV];
    const int pcv3 = pcv == PREC_F16 || pcv == PREC_F16X2 ? PREC_BF16X3 : pcv;  // the fp32-A and streaming-upsample kernels have no fp16 variant
    // re-derive the plan
    size_t max_img = 0, max_hid = 0;
    {
        long Ti = T;
        for (int i = 0; i < S; ++i) {
            if (i > 0) Ti *= s.ratios[i - 1];
            size_t C = (size_t)s.n_filters << (S - 1 - i);
            size_t img = (size_t)B * (pad + Ti) * C;
            max_img = img > max_img ? img : max_img;
            size_t hid = (size_t)B * Ti * C * s.ffn_mult;
            max_hid = hid > max_hid ? hid : max_hid;
        }
    }
    Bump bump(ws);
    CodecWs w;
    w.plan(bump, max_img, max_img, max_hid, (size_t)B * (pad + T) * L);

    // latent image with causal zero pad, then stem conv k (as GEMM over K*latent contiguous floats)
    HIPC(launch_zero_pad_frames(w.lat, B, T, L, pad, st));
    HIPC(hipMemcpy2DAsync(w.lat + (long)pad * L, (size_t)(pad + T) * L * 4, latents, (size_t)T * L * 4, (size_t)T * L * 4, B,
                          hipMemcpyDeviceToDevice, st));
    float* x = w.xa;
    float* xn = w.xb;
    int Ti = T;
    int C = dec_.stages[0].C;
    HIPC(launch_zero_pad_frames3(x, xn, w.nb, B, Ti, C, pad, st));
    {
        RowMap am = rowmap_batched(L, Ti, (long)(pad + Ti) * L, (long)(pad - (Kc - 1)) * L);
        RowMap om = rowmap_batched(C, Ti, (long)(pad + Ti) * C, (long)pad * C);
        HIPC(gemm_store(ops(w.lat, am, dec_.stem, B * Ti), ACT_NONE, store_to(x, om, dec_.stem_b), 1, pcv3, st));
    }
    static const char* kDecTags[] = {"dec.s0", "dec.s1", "dec.s2", "dec.s3", "dec.s4", "dec.s5", "dec.s6", "dec.s7"};
    for (int i = 0; i < S; ++i) {
        ProfTag ptag(kDecTags[i < 8 ? i : 7]);
        const CodecStageW& sg = dec_.stages[i];
        if (i > 0) {
            // ConvTranspose1d(k = 2r, stride r), causal trim: rows (x[t-1], x[t]) -> r output frames
            const int r = sg.r, Cn = sg.C, Tn = Ti * r;
            RowMap am = rowmap_batched(C, Ti, (long)(pad + Ti) * C, (long)(pad - 1) * C);
            RowMap om = rowmap_batched((long)r * Cn, Ti, (long)(pad + Tn) * Cn, (long)pad * Cn);
            if (fused_ffn_ && codec_upsample_wave_ok(sg.resample.K, sg.resample.N) && sg.resample.K == 2 * C && sg.resample.N == r * Cn)
                HIPC(launch_codec_upsample_wave(x, am, sg.resample.hi, sg.resample.lo, sg.resample.K, sg.resample_bias, xn, om,
                                                B * Ti, sg.resample.K, sg.resample.N, pcv3, st));
            else if (pcv == PREC_F16X2 && sg.resample.l16 && sg.resample.K % 64 == 0 && sg.resample.K >= x2_mink_ && sg.resample.K <= x2_maxk_ &&
                     C % 8 == 0 && (size_t)B * (pad + Ti) * C <= max_img) {
                // two-pass fp16 product: the image once as ONE fp16 array (pads included: the causal zeros), the weights as an fp16
                // hi + lo pair — A W_lo + A W_hi on the DMA-ring GEMM instead of three split-bf16 passes on the fp32-A kernel
                HIPC(launch_to_split(x, rowmap_plain(C), w.n2hi, sm_lo_for(PREC_F16, nullptr, satp(SITE_CODEC_CONV)), rowmap_plain(C), B * (pad + Ti), C, st));
                Gemm3Operands g3 = ops3(SplitBuf{w.n2hi, w.n2lo}, am, sg.resample, B * Ti, PREC_F16);
                g3.Wlo = sg.resample.l16;
                HIPC(gemm3_store_x2(g3, store_to(xn, om, sg.resample_bias), st));
            } else if (fused_ffn_ && sg.resample.K % 64 == 0 && sg.resample.K >= up_g3_mink_ && C % 8 == 0 && (size_t)B * (pad + Ti) * C <= max_img) {
                const int pg = pcv == PREC_F16X2 ? PREC_BF16X3 : pcv;   // (f16x2 outside its K range: split-bf16)
                // widest stages (K >= 2048; measured: 215 -> 148 us and 216 -> 193 us, no gain at K <= 1024): split the image once (pads included: they are the causal zeros) and run the DMA-ring GEMM on
                // the overlapping rows of the split pair (n2 is free between blocks)
                SplitBuf xs{w.n2hi, w.n2lo};
                HIPC(launch_to_split(x, rowmap_plain(C), xs.hi, xs.as(pg, satp(SITE_CODEC_CONV)).lo, rowmap_plain(C), B * (pad + Ti), C, st));
                HIPC(gemm3_store(ops3(xs, am, sg.resample, B * Ti, pg), ACT_NONE, store_to(xn, om, sg.resample_bias), 1, pg, st));
            } else
                HIPC(gemm_store(ops(x, am, sg.resample, B * Ti), ACT_NONE, store_to(xn, om, sg.resample_bias), 1, pcv3, st));
            float* t = x; x = xn; xn = t;
            Ti = Tn;
            C = Cn;
            // zero pads of the three images at the new geometry, in one launch BEHIND the product: it writes data rows only, and its
            // input (now the ping-pong partner) is free to be overwritten from here on
            HIPC(launch_zero_pad_frames3(x, xn, w.nb, B, Ti, C, pad, st));
        }
        if (const int ch = codec_stage_chain(st, sg, &x, &xn, B, Ti, C)) {   // > 0: the whole stage went out as one launch
            if (ch < 0) return 1;
            continue;
        }
        for (const CodecBlockW& b : sg.blocks)
            if (codec_block(st, b, &x, &xn, w.nb, w.n2hi, w.n2lo, w.hhi, w.hlo, B, Ti, C, max_img)) return 1;
    }
    if (dec_.final_norm_w) {   // out of place into the (zero-padded) scratch image: the head conv reads K - 1 pad frames
        const RowMap img = rowmap_batched(C, Ti, (long)(pad + Ti) * C, (long)pad * C);
        HIPC(launch_rmsnorm(x, img, w.nb, nullptr, nullptr, img, B * Ti, C, cspec_.eps, dec_.final_norm_w, st));
        x = w.nb;
    }
    HIPC(launch_head_conv(x, dec_.head_w, dec_.head_b_host, audio, B, Ti, C, Kc, pad, st));
    return 0;
}

size_t Engine::encode_ws_bytes(int B, int S_) const {
    const CodecSpecC& s = cspec_;
    const int S = s.n_ratios + 1;
    size_t max_img = 0, max_hid = 0;
    long Ti = S_;
    for (int i = 0; i < S; ++i) {
        if (i > 0) Ti /= s.ratios[s.n_ratios - i];
        size_t C = (size_t)s.n_filters << i;
        size_t img = (size_t)B * (kCodecPad + Ti) * C;
        max_img = img > max_img ? img : max_img;
        size_t hid = (size_t)B * Ti * C * s.ffn_mult;
        max_hid = hid > max_hid ? hid : max_hid;
    }
    Bump b(nullptr);
    CodecWs w;
    w.plan(b, max_img, max_img, max_hid, 1);
    return b.off + 256;
}

int Engine::codec_encode(hipStream_t st, const float* audio, int B, int S_, float* latents, void* ws, size_t ws_bytes) {
    DeepScope deep_scope(gemm_deep_);
    PersistScope persist_scope(persist_cus_);
    if (!enc_.ready) return fail("codec_encode: encoder weights not finalized");
    if (ws_bytes < encode_ws_bytes(B, S_)) return fail("codec_encode: workspace too small");
    HIPC(hipSetDevice(device_));
    const CodecSpecC& s = cspec_;
    const int S = s.n_ratios + 1, pad = kCodecPad, Kc = s.kernel;
    const int pcv3 = prec_[SITE_CODEC_CONV] == PREC_F16 || prec_[SITE_CODEC_CONV] == PREC_F16X2 ? PREC_BF16X3 : prec_[SITE_CODEC_CONV];
    size_t max_img = 0, max_hid = 0;
    {
        long Ti = S_;
        for (int i = 0; i < S; ++i) {
            if (i > 0) Ti /= s.ratios[s.n_ratios - i];
            size_t C = (size_t)s.n_filters << i;
            size_t img = (size_t)B * (pad + Ti) * C;
            max_img = img > max_img ? img : max_img;
            size_t hid = (size_t)B * Ti * C * s.ffn_mult;
            max_hid = hid > max_hid ? hid : max_hid;
        }
    }
    Bump bump(ws);
    CodecWs w;
    w.plan(bump, max_img, max_img, max_hid, 1);
    float* x = w.xa;
    float* xn = w.xb;
    int Ti = S_;
    int C = enc_.stages[0].C;
    // The coarse end of the encoder on short references is a handful of rows against a long K (2-s references: the last strided
    // conv is 120 x 2048 x 16384, the head 120 x 64 x 14336): as row tiles alone that is 2-32 workgroups streaming 134 MB of
    // weights (374 / 254 us).  There: convert the (small) image once to the GEMM operand format — pads included, they are the
    // causal zeros — and run gemm3 over the overlapping rows as K slices + one reduce.  Returns true when it took the product.
    bool conv_err = false;
    const int pcv = prec_[SITE_CODEC_CONV];
    auto conv_small_m = [&](const float* img, long img_rows, int Cin, const RowMap& am, const PW& wt, const float* bias, float* out,
                            const RowMap& om, int M) -> bool {
        const int splits = (fused_ffn_ && wt.K >= 2048) ? small_m_splits(M, wt.K, 8, 4) : 1;
        if (splits < 2 || wt.K % 64 || Cin % 4 || wt.N % 4 || (size_t)img_rows * Cin > max_img || (size_t)splits * M * wt.N * 2 > max_hid)
            return false;
        SplitBuf xs{w.n2hi, w.n2lo};
        float* part = reinterpret_cast<float*>(w.hhi);   // max_hid bf16 = max_hid / 2 floats
        hipError_t e = launch_to_split(img, rowmap_plain(Cin), xs.hi, xs.as(pcv, satp(SITE_CODEC_CONV)).lo, rowmap_plain(Cin), (int)img_rows, Cin, st);
        if (e == hipSuccess) e = gemm3_store_splitk(ops3(xs, am, wt, M, pcv), part, splits, pcv, bias, out, om, st);
        if (e != hipSuccess) { fail_hip(e, "codec_encode: split-K conv"); conv_err = true; }
        return true;
    };
    HIPC(launch_zero_pad_frames3(x, xn, w.nb, B, Ti, C, pad, st));
    HIPC(launch_stem_conv1(audio, enc_.stem_w_raw, enc_.stem_b, x, B, Ti, C, Kc, pad, st));
    static const char* kEncTags[] = {"cenc.s0", "cenc.s1", "cenc.s2", "cenc.s3", "cenc.s4", "cenc.s5", "cenc.s6", "cenc.s7"};
    for (int i = 0; i < S; ++i) {
        ProfTag ptag(kEncTags[i < 8 ? i : 7]);
        const CodecStageW& sg = enc_.stages[i];
        if (i > 0) {
            // Conv1d(k = 2r, stride r), causal left pad r: out[t] reads frames [(t-1) r, (t+1) r)
            const int r = sg.r, Cn = sg.C, Tn = Ti / r;
            RowMap am = rowmap_batched((long)r * C, Tn, (long)(pad + Ti) * C, (long)(pad - r) * C);
            RowMap om = rowmap_batched(Cn, Tn, (long)(pad + Tn) * Cn, (long)pad * Cn);
            if (!conv_small_m(x, (long)B * (pad + Ti), C, am, sg.resample, sg.resample_bias, xn, om, B * Tn))
                HIPC(gemm_store(ops(x, am, sg.resample, B * Tn), ACT_NONE, store_to(xn, om, sg.resample_bias), 1, pcv3, st));
            if (conv_err) return 1;
            float* t = x; x = xn; xn = t;
            Ti = Tn;
            C = Cn;
            HIPC(launch_zero_pad_frames3(x, xn, w.nb, B, Ti, C, pad, st));   // (behind the product, as in the decoder)
        }
        if (const int ch = codec_stage_chain(st, sg, &x, &xn, B, Ti, C)) {
            if (ch < 0) return 1;
            continue;
        }
        for (const CodecBlockW& b : sg.blocks)
            if (codec_block(st, b, &x, &xn, w.nb, w.n2hi, w.n2lo, w.hhi, w.hlo, B, Ti, C, max_img)) return 1;
    }
    if (enc_.final_norm_w) {
        const RowMap img = rowmap_batched(C, Ti, (long)(pad + Ti) * C, (long)pad * C);
        HIPC(launch_rmsnorm(x, img, w.nb, nullptr, nullptr, img, B * Ti, C, cspec_.eps, enc_.final_norm_w, st));
        x = w.nb;
    }
    RowMap am = rowmap_batched(C, Ti, (long)(pad + Ti) * C, (long)(pad - (Kc - 1)) * C);
    if (!conv_small_m(x, (long)B * (pad + Ti), C, am, enc_.head, enc_.head_b, latents, rowmap_plain(s.latent_dim), B * Ti))
        HIPC(gemm_store(ops(x, am, enc_.head, B * Ti), ACT_NONE, store_to(latents, rowmap_plain(s.latent_dim), enc_.head_b), 1,
                        pcv3, st));
    return conv_err ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// test hooks
// ---------------------------------------------------------------------------------------------
int Engine::test_gemm(hipStream_t st, const float* A, int lda, const float* W, const float* bias, int M, int N, int K,
                      int act, int split, int cfg, float* C, int ldc) {
    DeepScope deep_scope(gemm_deep_);
    HIPC(hipSetDevice(device_));
    bf16_t *hi = nullptr, *lo = nullptr;
    HIPC(hipMalloc(&hi, (size_t)N * K * 2));
    HIPC(hipMalloc(&lo, (size_t)N * K * 2));
    HIPC(launch_split_rows(W, K, hi, lo, K, N, K, nullptr, st));
    PW w;
    w.hi = hi; w.lo = lo; w.N = N; w.K = K;
    hipError_t e = gemm_store(ops(A, rowmap_plain(lda), w, M), act, store_to(C, rowmap_plain(ldc), bias), 1, split, st, cfg);
    (void)hipStreamSynchronize(st);
    (void)hipFree(hi);
    (void)hipFree(lo);
    return e == hipSuccess ? 0 : fail_hip(e, "test_gemm");
}

int Engine::test_gemm3(hipStream_t st, const float* A, const float* W, const float* bias, int M, int N, int K, int act,
                       int split, int cfg, float* C) {
    DeepScope deep_scope(gemm_deep_);
    HIPC(hipSetDevice(device_));
    bf16_t *hi = nullptr, *lo = nullptr, *ahi = nullptr, *alo = nullptr;
    HIPC(hipMalloc(&hi, (size_t)N * K * 2));
    HIPC(hipMalloc(&lo, (size_t)N * K * 2));
    HIPC(hipMalloc(&ahi, (size_t)M * K * 2));
    HIPC(hipMalloc(&alo, (size_t)M * K * 2));
    // PREC_F16: the single 16-bit arrays (hi) carry fp16
    HIPC(launch_split_rows(W, K, split == PREC_F16 ? nullptr : hi, lo, K, N, K, nullptr, st, split == PREC_F16 ? hi : nullptr));
    HIPC(launch_to_split(A, rowmap_plain(K), ahi, sm_lo_for(split, alo), rowmap_plain(K), M, K, st));
    PW w;
    w.hi = hi; w.lo = lo; w.h16 = hi; w.N = N; w.K = K;
    hipError_t e = gemm3_store(ops3(SplitBuf{ahi, alo}, rowmap_plain(K), w, M, split), act, store_to(C, rowmap_plain(N), bias), 1,
                               split, st, cfg);
    (void)hipStreamSynchronize(st);
    for (void* p : {(void*)hi, (void*)lo, (void*)ahi, (void*)alo}) (void)hipFree(p);
    return e == hipSuccess ? 0 : fail_hip(e, "test_gemm3");
}

int Engine::test_swiglu(hipStream_t st, const float* A, const float* W1, const float* W3, const float* b1,
                        const float* b3, int M, int F, int K, int split, float* out) {
    DeepScope deep_scope(gemm_deep_);
    HIPC(hipSetDevice(device_));
    if (F % 32) return fail("test_swiglu: F must be a multiple of 32");
    float* cat = nullptr;
    int* dperm = nullptr;
    bf16_t *hi = nullptr, *lo = nullptr;
    std::vector<int> perm = swiglu_perm(F);
    HIPC(hipMalloc(&cat, (size_t)2 * F * K * 4));
    HIPC(hipMalloc(&dperm, perm.size() * sizeof(int)));
    HIPC(hipMalloc(&hi, (size_t)2 * F * K * 2));
    HIPC(hipMalloc(&lo, (size_t)2 * F * K * 2));
    HIPC(hipMemcpyAsync(cat, W1, (size_t)F * K * 4, hipMemcpyDeviceToDevice, st));
    HIPC(hipMemcpyAsync(cat + (size_t)F * K, W3, (size_t)F * K * 4, hipMemcpyDeviceToDevice, st));
    HIPC(hipMemcpyAsync(dperm, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPC(launch_split_rows(cat, K, hi, lo, K, 2 * F, K, dperm, st));
    PW w;
    w.hi = hi; w.lo = lo; w.N = 2 * F; w.K = K;
    EpiSwiGLU sw{out, F, b1, b3, nullptr, nullptr};
#ifdef SMTTS_TEST_KERNELS
    hipError_t e = gemm_swiglu(ops(A, rowmap_plain(K), w, M), sw, split, st);
#else
    hipError_t e = hipErrorNotSupported;   // (the v1 SwiGLU instantiations are test kernels: make TEST_KERNELS=1)
#endif
    (void)hipStreamSynchronize(st);
    (void)hipFree(cat); (void)hipFree(dperm); (void)hipFree(hi); (void)hipFree(lo);
    return e == hipSuccess ? 0 : fail_hip(e, "test_swiglu");
}

// epi: 0 store, 1 store+gelu, 2 swiglu (N = packed 2F), 3 resid tanh-gate
int Engine::bench_gemm(int M, int N, int K, int epi, int split, int cfg, int iters, int ver, float* avg_us) {
    DeepScope deep_scope(gemm_deep_);
    HIPC(hipSetDevice(device_));
    float *A = nullptr, *Wf = nullptr, *C = nullptr, *bias = nullptr, *gate = nullptr;
    bf16_t *hi = nullptr, *lo = nullptr, *ahi = nullptr, *alo = nullptr;
    HIPC(hipMalloc(&ahi, (size_t)M * K * 2));
    HIPC(hipMalloc(&alo, (size_t)M * K * 2));
    HIPC(hipMalloc(&A, (size_t)M * K * 4));
    HIPC(hipMalloc(&Wf, (size_t)N * K * 4));
    HIPC(hipMalloc(&C, (size_t)M * N * 4));
    HIPC(hipMalloc(&bias, (size_t)N * 4));
    HIPC(hipMalloc(&gate, (size_t)N * 4));
    HIPC(hipMalloc(&hi, (size_t)N * K * 2));
    HIPC(hipMalloc(&lo, (size_t)N * K * 2));
    uint8_t* maskb = nullptr;
    float* partb = nullptr;
    HIPC(hipMalloc(&maskb, (size_t)M));
    bf16_t* yimg = nullptr;
    HIPC(hipMalloc(&yimg, (size_t)M * N * 2));
    HIPC(hipMalloc(&partb, (size_t)M * ((N + 31) / 32) * 2 * 4));
    HIPC(hipMemset(maskb, 1, (size_t)M));
    HIPC(launch_synth(A, (long)M * K, 1, 0.f, 1.f, 0));
    HIPC(launch_synth(Wf, (long)N * K, 2, 0.f, 0.05f, 0));
    HIPC(launch_synth(bias, N, 3, 0.f, 0.1f, 0));
    HIPC(launch_synth(gate, N, 4, 0.f, 0.5f, 0));
    HIPC(hipMemsetAsync(C, 0, (size_t)M * N * 4, 0));
    HIPC(launch_split_rows(Wf, K, split == PREC_F16 ? nullptr : hi, lo, K, N, K, nullptr, 0, split == PREC_F16 ? hi : nullptr));
    PW w;
    w.hi = hi; w.lo = lo; w.h16 = hi; w.N = N; w.K = K;
    GemmOperands g = ops(A, rowmap_plain(K), w, M);
    HIPC(launch_to_split(A, rowmap_plain(K), ahi, sm_lo_for(split, alo), rowmap_plain(K), M, K, 0));
    Gemm3Operands g3 = ops3(SplitBuf{ahi, alo}, rowmap_plain(K), w, M, split);
    auto run = [&]() -> hipError_t {
        if (ver == 3) {
            switch (epi) {
                case 0: return gemm3_store(g3, ACT_NONE, store_to(C, rowmap_plain(N), bias), 1, split, 0, cfg);
                case 1: return gemm3_store(g3, ACT_GELU, store_to(C, rowmap_plain(N), bias), 1, split, 0, cfg);
                case 2: { EpiSwiGLU sw{C, N / 2, bias, bias, nullptr, nullptr}; return gemm3_swiglu(g3, sw, split, 0); }
                case 4: {   // GELU hidden written as ONE 16-bit operand array (the codec's first FFN product): C is reused as that array
                    EpiStore<ACT_NONE> e{nullptr, rowmap_plain(N), 0, bias, 0, 1.f, nullptr, reinterpret_cast<bf16_t*>(C), sm_lo_for(split == PREC_BF16X3 ? PREC_F16 : split, nullptr)};
                    return gemm3_store(g3, ACT_GELU, e, 1, split, 0, cfg);
                }
                case 5: { EpiResid<0> r{C, rowmap_plain(N), bias, gate, 0, 0, 0, M, nullptr}; return gemm3_resid(g3, 2, r, split, 0, cfg); }   // LayerScale residual (codec FF2)
                case 6: return gemm3_store(g3, ACT_NONE, store_to(C, rowmap_plain(N), nullptr), 1, split, 0, cfg);   // no bias: what a split-K slice stores
                case 7: case 9: {   // LN-fold producer (the DiT's out-proj with a row mask / FF2 without): residual + next operand image + row partials
                    EpiResidLN r{C, rowmap_plain(N), bias, gate, epi == 7 ? maskb : nullptr, bias, yimg, sm_lo_for(split == PREC_BF16X3 ? PREC_F16 : split, nullptr), N, partb, N / 32};
                    return gemm3_resid_ln(g3, r, split, 0, cfg);
                }
                case 8: { EpiResid<0> r{C, rowmap_plain(N), bias, gate, 0, 0, 0, M, maskb}; return gemm3_resid(g3, 1, r, split, 0, cfg); }   // tanh-gated residual with a row mask
                default: { EpiResid<0> r{C, rowmap_plain(N), bias, gate, 0, 0, 0, M, nullptr}; return gemm3_resid(g3, 1, r, split, 0, cfg); }
            }
        }
        switch (epi) {
            case 0: return gemm_store(g, ACT_NONE, store_to(C, rowmap_plain(N), bias), 1, split, 0, cfg);
            case 1: return gemm_store(g, ACT_GELU, store_to(C, rowmap_plain(N), bias), 1, split, 0, cfg);
#ifdef SMTTS_TEST_KERNELS
            case 2: { EpiSwiGLU sw{C, N / 2, bias, bias, nullptr, nullptr}; return gemm_swiglu(g, sw, split, 0); }
#endif
            default: { EpiResid<0> r{C, rowmap_plain(N), bias, gate, 0, 0, 0, M, nullptr}; return gemm_resid(g, 1, r, split, 0, cfg); }
        }
    };
    for (int i = 0; i < 3; ++i) HIPC(run());
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) HIPC(run());
    HIPC(hipEventRecord(e1, 0));
    HIPC(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.f / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    for (void* p : {(void*)A, (void*)Wf, (void*)C, (void*)bias, (void*)gate, (void*)hi, (void*)lo, (void*)ahi, (void*)alo, (void*)maskb, (void*)partb, (void*)yimg})
        (void)hipFree(p);
    return 0;
}
