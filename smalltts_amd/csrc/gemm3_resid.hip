#include "gemm_ops.hpp"
#include "prof.hpp"
template <int G>
static EpiResid<G> conv(const EpiResid<0>& p) {
    return EpiResid<G>{p.x, p.xmap, p.bias, p.gate, p.gld, p.grow0, p.grstride, p.rows_per_batch, p.rowmask};
}
hipError_t gemm3_resid(const Gemm3Operands& g, int gate_mode, const EpiResid<0>& p, int split, hipStream_t st, int cfg) {
    static const char* names[] = {"resid", "resid_gate", "resid_layerscale"};
    ProfScope ps(st, gemm3_prof_name(g, false, cfg, split, names[gate_mode % 3]), gemm3_flops(g, 1), gemm3_bytes(g, 1, split, 8.0), gemm_bytes8d(g.N, g.K, (g.ksplit_tiles ? 1 : 1)));
    switch (gate_mode) {
        case 0: return gemm3_launch(g, p, 1, split, st, cfg);
        case 1: return gemm3_launch(g, conv<1>(p), 1, split, st, cfg);
        case 2: return gemm3_launch(g, conv<2>(p), 1, split, st, cfg);
    }
    return hipErrorInvalidValue;
}
