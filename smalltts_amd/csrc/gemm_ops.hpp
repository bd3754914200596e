// Non-template entry points for the fused GEMM family (definitions in gemm_store.hip,
// gemm_resid.hip, gemm_misc.hip so the instantiations compile in parallel).
#pragma once
#include <string>

#include "gemm3.hpp"

// kernel identity used by the profiler, e.g. "gemm<64x128x64,s3,store_gelu>"; mirrors the template
// arguments of the gemm_kernel instantiation that gemm_launch() picks for (cfg, split, K).
static inline std::string gemm_prof_name(const GemmOperands& g, bool paired, int cfg, int split, const char* epi) {
    if (cfg < 0) cfg = gemm_pick_cfg(g.M, g.N, g.K, paired);
    static const char* tiles[] = {"64x128x64", "64x64x64", "128x128x64", "128x32x64", "128x64x64"};
    std::string t = tiles[cfg];
    if (cfg == CFG_128x64 && g.K <= 32) t = "128x64x32";
    return std::string("gemm<") + t + ",s" + std::to_string(split) + "," + epi + ">";
}
// algorithmic work of one GEMM launch: 2*M*N*K flops (x Z); bytes = A fp32 read once + W (bf16 hi[+lo]) read
// once + C fp32 written once (+ read once for residual epilogues)
static inline double gemm_flops(const GemmOperands& g, int Z) { return 2.0 * g.M * (double)g.N * g.K * Z; }
static inline double gemm_bytes(const GemmOperands& g, int Z, int split, double c_elems_per_out, bool w_shared = false) {
    double w = (double)g.N * g.K * 2.0 * (split == 3 ? 2 : 1) * (w_shared ? 1 : Z);
    return Z * ((double)g.M * g.K * 4.0 + (double)g.M * g.N * 4.0 * c_elems_per_out) + w;
}

// act: ACT_NONE / ACT_SILU / ACT_GELU / ACT_MISH
hipError_t gemm_store(const GemmOperands& g, int act, const EpiStore<ACT_NONE>& p, int Z, int split, hipStream_t st, int cfg = -1);
hipError_t gemm_swiglu(const GemmOperands& g, const EpiSwiGLU& p, int split, hipStream_t st);
// gate_mode 0/1/2 (see EpiResid)
hipError_t gemm_resid(const GemmOperands& g, int gate_mode, const EpiResid<0>& p, int split, hipStream_t st, int cfg = -1);
hipError_t gemm_kv(const GemmOperands& g, const EpiKV& p, int split, hipStream_t st);
hipError_t gemm_convpos(const GemmOperands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st);

// ---- v3 (split-A, DMA ring, 8 waves) entry points: definitions in gemm3_ops.hip --------------------------
static inline std::string gemm3_prof_name(const Gemm3Operands& g, bool paired, int cfg, int split, const char* epi) {
    if (cfg < 0) cfg = gemm3_pick_cfg(g.M, g.N, paired, split != PREC_BF16X3);
    static const char* tiles[] = {"64x128", "128x128", "64x64", "128x64", "128x32", "160x128", "128x128w4", "256x256", "64x32", "32x64"};
    if (cfg == G4_256x256 && (split == PREC_BF16X3 || !gemm4_ok(g))) cfg = G3_128x128;   // (gemm3_launch's fallback)
    std::string n = std::string(cfg == G4_256x256 ? "gemm4<" : "gemm3<") + tiles[cfg] + ",s" + std::to_string(split) + "," + epi + ">";
    extern thread_local int g_prof_shapes;   // profile mode 3: the product's shape behind the class name ("... 600x3840x960[/k3]")
    if (g_prof_shapes) {
        n += " " + std::to_string(g.M) + "x" + std::to_string(g.N) + "x" + std::to_string(g.K);
        if (g.ksplit_tiles) n += "/k" + std::to_string((g.K / 64 + g.ksplit_tiles - 1) / g.ksplit_tiles);
    }
    return n;
}
// split-K launches (ksplit_tiles > 0) use blockIdx.z for K slices of ONE product: the work is counted once
static inline double gemm3_flops(const Gemm3Operands& g, int Z) {
    return 2.0 * g.M * (double)g.N * g.K * (g.ksplit_tiles ? 1 : Z);
}
static inline double gemm3_bytes(const Gemm3Operands& g, int Z, int split, double c_bytes_per_out, bool w_shared = false) {
    const double e = split == 3 ? 4.0 : 2.0;
    if (g.ksplit_tiles)  // operands once, Z fp32 partial outputs
        return (double)g.M * g.K * e + (double)g.N * g.K * e + Z * (double)g.M * g.N * c_bytes_per_out;
    return Z * ((double)g.M * g.K * e + (double)g.M * g.N * c_bytes_per_out) + (double)g.N * g.K * e * (w_shared ? 1 : Z);
}
// SURVEY 8(d) bytes of a GEMM launch: the weights once at 2 B / parameter ("bf16 weights read once per use, activations negligible")
static inline double gemm_bytes8d(int N, int K, int Z, bool w_shared = false) { return (double)N * K * 2.0 * (w_shared ? 1 : Z); }
hipError_t gemm3_store(const Gemm3Operands& g, int act, const EpiStore<ACT_NONE>& p, int Z, int split, hipStream_t st, int cfg = -1);
hipError_t gemm3_store_x2(const Gemm3Operands& g, const EpiStore<ACT_NONE>& p, hipStream_t st, int cfg = -1);   // PREC_F16X2: A = g.Ahi (fp16), W = g.Whi + g.Wlo (fp16 pair)
hipError_t gemm3_swiglu(const Gemm3Operands& g, const EpiSwiGLU& p, int split, hipStream_t st);
hipError_t gemm3_resid(const Gemm3Operands& g, int gate_mode, const EpiResid<0>& p, int split, hipStream_t st, int cfg = -1);
hipError_t gemm3_resid_ln(const Gemm3Operands& g, const EpiResidLN& p, int split, hipStream_t st, int cfg = -1);   // LN-fold producer (N % 32 == 0)
hipError_t gemm3_kv(const Gemm3Operands& g, const EpiKV& p, int split, hipStream_t st);
hipError_t gemm3_convpos(const Gemm3Operands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st);
// QKVG projection -> attention operand images (EpiQKV): columns n = (part * H + h) * HW + d, N = 4 * H * HW
hipError_t gemm3_qkv(const Gemm3Operands& g, const EpiQKV& p, int split, hipStream_t st);
