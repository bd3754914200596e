#include "gemm_ops.hpp"
#include "prof.hpp"
template <int ACT>
static EpiStore<ACT> conv(const EpiStore<ACT_NONE>& p) {
    return EpiStore<ACT>{p.out, p.omap, p.o_z, p.bias, p.bias_z, p.scale, p.rowmask, p.ohi, p.olo};
}
hipError_t gemm_store(const GemmOperands& g, int act, const EpiStore<ACT_NONE>& p, int Z, int split, hipStream_t st, int cfg) {
    static const char* names[] = {"store", "store_silu", "store_gelu", "store_mish"};
    ProfScope ps(st, gemm_prof_name(g, false, cfg, split, names[act & 3]), gemm_flops(g, Z),
                 gemm_bytes(g, Z, split, 1.0), gemm_bytes8d(g.N, g.K, Z));
    switch (act) {
        case ACT_NONE: return gemm_launch(g, p, Z, split, st, cfg);
        case ACT_SILU: return gemm_launch(g, conv<ACT_SILU>(p), Z, split, st, cfg);
        case ACT_GELU: return gemm_launch(g, conv<ACT_GELU>(p), Z, split, st, cfg);
        case ACT_MISH: return gemm_launch(g, conv<ACT_MISH>(p), Z, split, st, cfg);
    }
    return hipErrorInvalidValue;
}
