#include "gemm_ops.hpp"
hipError_t gemm_swiglu(const GemmOperands& g, const EpiSwiGLU& p, int split, hipStream_t st) {
    return gemm_launch(g, p, 1, split, st, CFG_64x128);
}
hipError_t gemm_kv(const GemmOperands& g, const EpiKV& p, int split, hipStream_t st) {
    return gemm_launch(g, p, 1, split, st, -1);
}
hipError_t gemm_convpos(const GemmOperands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st) {
    if (final) {
        EpiConvPos<1> q{p.out, p.h, p.bias, p.mask, p.G, p.cpg, p.T, p.pad, p.gstride};
        return gemm_launch(g, q, Z, split, st, CFG_64x64);
    }
    return gemm_launch(g, p, Z, split, st, CFG_64x64);
}
