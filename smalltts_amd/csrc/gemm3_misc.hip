#include "gemm_ops.hpp"
#include "prof.hpp"
hipError_t gemm3_swiglu(const Gemm3Operands& g, const EpiSwiGLU& p, int split, hipStream_t st) {
    const int cfg = gemm3_pick_cfg(g.M, g.N, true, split != PREC_BF16X3);
    ProfScope ps(st, gemm3_prof_name(g, true, cfg, split, "swiglu"), gemm3_flops(g, 1), gemm3_bytes(g, 1, split, 2.0), gemm_bytes8d(g.N, g.K, (g.ksplit_tiles ? 1 : 1)));
    if (p.fold.part) {   // LN-fold consumer: its own instantiations
        const EpiSwiGLUFold q{p.out, p.ldo, p.b1, p.b3, p.ohi, p.olo, p.fold};
        return gemm3_launch(g, q, 1, split, st, cfg);
    }
    return gemm3_launch(g, p, 1, split, st, cfg);
}
hipError_t gemm3_kv(const Gemm3Operands& g, const EpiKV& p, int split, hipStream_t st) {
    ProfScope ps(st, gemm3_prof_name(g, false, -1, split, "kv_scatter"), gemm3_flops(g, 1), gemm3_bytes(g, 1, split, 4.0), gemm_bytes8d(g.N, g.K, (g.ksplit_tiles ? 1 : 1)));
    return gemm3_launch(g, p, 1, split, st, -1);
}
hipError_t gemm3_convpos(const Gemm3Operands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st) {
    ProfScope ps(st, gemm3_prof_name(g, false, G3_64x64, split, final ? "convpos_final" : "convpos"), gemm3_flops(g, Z),
                 gemm3_bytes(g, Z, split, 4.0, true), gemm_bytes8d(g.N, g.K, (g.ksplit_tiles ? 1 : Z), true));
    if (final) {
        EpiConvPos<1> q{p.out, p.h, p.bias, p.mask, p.G, p.cpg, p.T, p.pad, p.gstride, nullptr, nullptr, p.by_group};
        return gemm3_launch(g, q, Z, split, st, G3_64x64);
    }
    return gemm3_launch(g, p, Z, split, st, G3_64x64);
}
