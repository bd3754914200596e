#include "gemm_ops.hpp"
#include "prof.hpp"
template <int G>
static EpiResid<G> conv(const EpiResid<0>& p) {
    return EpiResid<G>{p.x, p.xmap, p.bias, p.gate, p.gld, p.grow0, p.grstride, p.rows_per_batch, p.rowmask};
}
hipError_t gemm_resid(const GemmOperands& g, int gate_mode, const EpiResid<0>& p, int split, hipStream_t st, int cfg) {
    static const char* names[] = {"resid", "resid_tanhgate", "resid_layerscale"};
    ProfScope ps(st, gemm_prof_name(g, false, cfg, split, names[gate_mode % 3]), gemm_flops(g, 1),
                 gemm_bytes(g, 1, split, 2.0), gemm_bytes8d(g.N, g.K, 1));
    switch (gate_mode) {
        case 0: return gemm_launch(g, p, 1, split, st, cfg);
        case 1: return gemm_launch(g, conv<1>(p), 1, split, st, cfg);
        case 2: return gemm_launch(g, conv<2>(p), 1, split, st, cfg);
    }
    return hipErrorInvalidValue;
}
