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
// the DiT's gated-residual projections with the LN-fold producer epilogue (gemm.hpp EpiResidLN): the residual row, the next GEMM's
// operand image and the row partials in one pass — no split-K partials, no norm launch
hipError_t gemm3_resid_ln(const Gemm3Operands& g, const EpiResidLN& p, int split, hipStream_t st, int cfg) {
    if (g.N % 32 || p.NP != g.N / 32 || !p.part || !p.yhi || !p.nscale) return hipErrorInvalidValue;
    ProfScope ps(st, gemm3_prof_name(g, false, cfg, split, "resid_ln"), gemm3_flops(g, 1), gemm3_bytes(g, 1, split, 8.0 + (sm_is_split(p.ylo) ? 4.0 : 2.0)),
                 gemm_bytes8d(g.N, g.K, 1));
    return gemm3_launch(g, p, 1, split, st, cfg);
}

G3_TIMELINE_EXPORTS(_resid)   // (lab builds with -DG3_TIMELINE only: tools/gemm3_resid_timeline.py)
