// QKVG projection with the attention-operand epilogue (gemm.hpp EpiQKV): 128-column tiles = whole (padded) heads.
#include <cstdlib>

#include "gemm_ops.hpp"
#include "prof.hpp"

template <int SPLIT, class E>
static hipError_t qkv_go(const Gemm3Operands& g, const E& p, bool big, bool deep, hipStream_t st) {
    if (big) {   // many rows (the teacher's 3B-row CFG batches): 128x128 tiles, 32x64 wave tiles
        if (deep) return gemm3_launch_cfg<128, 128, 4, 2, SPLIT, SPLIT == 3 ? 2 : 4, E>(g, p, 1, st);
        return gemm3_launch_cfg<128, 128, 4, 2, SPLIT, 2, E>(g, p, 1, st);
    }
    if (deep) return gemm3_launch_cfg<64, 128, 2, 4, SPLIT, SPLIT == 3 ? 3 : 6, E>(g, p, 1, st);
    return gemm3_launch_cfg<64, 128, 2, 4, SPLIT, 3, E>(g, p, 1, st);
}
template <class E>
static hipError_t qkv_split(const Gemm3Operands& g, const E& p, int split, bool big, bool deep, hipStream_t st) {
    if (split == PREC_BF16X3) return qkv_go<3>(g, p, big, deep, st);
    if (split == PREC_F16) return qkv_go<2>(g, p, big, deep, st);
    return qkv_go<1>(g, p, big, deep, st);
}

hipError_t gemm3_qkv(const Gemm3Operands& g_in, const EpiQKV& p, int split, hipStream_t st) {
    if (g_in.M <= 0) return hipSuccess;
    if (!gemm3_ok(g_in) || g_in.N != 4 * p.H * p.HW || (p.HW != 64 && p.HW != 128) || (p.dh & 1) || p.dh > p.HW || (p.rot_dim & 1) ||
        p.rot_dim > p.dh || (g_in.N % 128))
        return hipErrorInvalidValue;
    extern thread_local int g_gemm3_deep;
    extern int g_gemm3_nfast;
    Gemm3Operands g = g_in;
    g.nfast = g_gemm3_nfast && (long)g.M > (long)g.N;
    static const int big_minm = lab_env("SMTTS_QKV_BIG_MINM") ? atoi(lab_env("SMTTS_QKV_BIG_MINM")) : 641;   // A/B: 128x128 tiles from this many rows up
    const bool big = g.M >= big_minm;
    const long tiles = (long)((g.M + (big ? 127 : 63)) / (big ? 128 : 64)) * (g.N / 128);
    const bool deep = g_gemm3_deep && split != PREC_BF16X3 && tiles <= 256;   // deep rings only while the grid is one resident round (gemm3_launch_split)
    ProfScope ps(st, gemm3_prof_name(g, false, big ? G3_128x128 : G3_64x128, split, "qkv_img"), 2.0 * g.M * (double)g.N * g.K,
                 (split == 3 ? 4.0 : 2.0) * ((double)g.M * g.K + (double)g.N * g.K) + (p.prec == PREC_BF16X3 ? 4.0 : 2.0) * g.M * (double)g.N,
                 gemm_bytes8d(4 * p.H * p.dh, g.K, 1));
    if (p.fold.part) {   // LN-fold consumer: its own instantiations
        const EpiQKVFold q{p.bias, p.qw, p.kw, p.rope_cos, p.rope_sin, p.eps, p.q_scale, p.rot_dim, p.prec, p.q, p.q_lo, p.k, p.k_lo, p.vt, p.vt_lo,
                           p.g, p.g_lo, p.Nseq, p.H, p.dh, p.HW, p.Np, p.fold};
        return qkv_split(g, q, split, big, deep, st);
    }
    return qkv_split(g, p, split, big, deep, st);
}
