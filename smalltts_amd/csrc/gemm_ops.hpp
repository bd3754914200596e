// Non-template entry points for the fused GEMM family (definitions in gemm_store.hip,
// gemm_resid.hip, gemm_misc.hip so the instantiations compile in parallel).
#pragma once
#include "gemm.hpp"

// act: ACT_NONE / ACT_SILU / ACT_GELU / ACT_MISH
hipError_t gemm_store(const GemmOperands& g, int act, const EpiStore<ACT_NONE>& p, int Z, int split, hipStream_t st, int cfg = -1);
hipError_t gemm_swiglu(const GemmOperands& g, const EpiSwiGLU& p, int split, hipStream_t st);
// gate_mode 0/1/2 (see EpiResid)
hipError_t gemm_resid(const GemmOperands& g, int gate_mode, const EpiResid<0>& p, int split, hipStream_t st, int cfg = -1);
hipError_t gemm_kv(const GemmOperands& g, const EpiKV& p, int split, hipStream_t st);
hipError_t gemm_convpos(const GemmOperands& g, bool final, const EpiConvPos<0>& p, int Z, int split, hipStream_t st);
