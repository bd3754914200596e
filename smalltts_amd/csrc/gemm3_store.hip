#include "gemm_ops.hpp"
#include "prof.hpp"
template <int ACT>
static EpiStore<ACT> conv(const EpiStore<ACT_NONE>& p) {
    return EpiStore<ACT>{p.out, p.omap, p.o_z, p.bias, p.bias_z, p.scale, p.rowmask, p.ohi, p.olo};
}
hipError_t gemm3_store(const Gemm3Operands& g, int act, const EpiStore<ACT_NONE>& p, int Z, int split, hipStream_t st, int cfg) {
    static const char* names[] = {"store", "store_silu", "store_gelu", "store_mish"};
    ProfScope ps(st, gemm3_prof_name(g, false, cfg, split, names[act & 3]), gemm3_flops(g, Z), gemm3_bytes(g, Z, split, 4.0), gemm_bytes8d(g.N, g.K, (g.ksplit_tiles ? 1 : Z)));
    switch (act) {
        case ACT_NONE: return gemm3_launch(g, p, Z, split, st, cfg);
        case ACT_SILU: return gemm3_launch(g, conv<ACT_SILU>(p), Z, split, st, cfg);
        case ACT_GELU: return gemm3_launch(g, conv<ACT_GELU>(p), Z, split, st, cfg);
    }
    return hipErrorInvalidValue;
}

// PREC_F16X2 (gemm3_launch_x2): fp32 store epilogue only — the codec's ConvTranspose-as-GEMM products
hipError_t gemm3_store_x2(const Gemm3Operands& g, const EpiStore<ACT_NONE>& p, hipStream_t st, int cfg) {
    ProfScope ps(st, gemm3_prof_name(g, false, cfg < 0 ? gemm3_pick_cfg(g.M, g.N, false, false) : cfg, PREC_F16X2, "store"), gemm3_flops(g, 1),
                 (double)g.M * g.K * 2.0 + (double)g.N * g.K * 4.0 + (double)g.M * g.N * 4.0, gemm_bytes8d(g.N, g.K, 1));
    return gemm3_launch_x2(g, p, 1, st, cfg);
}

G3_TIMELINE_EXPORTS()   // (lab builds with -DG3_TIMELINE only: tools/gemm3_timeline.py)

#if defined(SMTTS_LAB) && defined(G4_TIMELINE)   // tools/gemm4_timeline.py
extern "C" int smtts_debug_read_timeline4(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4_tl_buf), (size_t)n * 8);
}
extern "C" int smtts_debug_clear_timeline4(void) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g4_tl_buf)) != hipSuccess) return 1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * (256 * 2 * 2 * 16 + 8));
}
#endif
