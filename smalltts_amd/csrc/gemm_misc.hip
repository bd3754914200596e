#include "gemm_ops.hpp"
#include "prof.hpp"
hipError_t gemm_swiglu(const GemmOperands& g, const EpiSwiGLU& p, int split, hipStream_t st) {
    ProfScope ps(st, gemm_prof_name(g, true, CFG_64x128, split, "swiglu"), gemm_flops(g, 1),
                 gemm_bytes(g, 1, split, 0.5), gemm_bytes8d(g.N, g.K, 1));
    return gemm_launch(g, p, 1, split, st, CFG_64x128);
}
hipError_t gemm_kv(const GemmOperands& g, const EpiKV& p, int split, hipStream_t st) {
    ProfScope ps(st, gemm_prof_name(g, false, -1, split, "kv_scatter"), gemm_flops(g, 1),
                 gemm_bytes(g, 1, split, 1.0), gemm_bytes8d(g.N, g.K, 1));
    return gemm_launch(g, p, 1, split, st, -1);
}
hipError_t gemm_convpos(const GemmOperands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st) {
    ProfScope ps(st, gemm_prof_name(g, false, CFG_64x64, split, final ? "convpos_final" : "convpos"),
                 gemm_flops(g, Z), gemm_bytes(g, Z, split, 1.0, true), gemm_bytes8d(g.N, g.K, Z, true));
    if (final) {
        EpiConvPos<1> q{p.out, p.h, p.bias, p.mask, p.G, p.cpg, p.T, p.pad, p.gstride, nullptr, nullptr, p.by_group};
        return gemm_launch(g, q, Z, split, st, CFG_64x64);
    }
    return gemm_launch(g, p, Z, split, st, CFG_64x64);
}
