// Lab-only instrumentation (never part of the shipped library): wave 0 of every workgroup stamps the shader clock at the markers the
// kernels carry (G3_STAMP / G3_STAMPK / G3_STAMPR in gemm3.hpp, EPI_STAMP in gemm.hpp's residual epilogues).  In the shipped build the
// markers are empty statements (gemm.hpp); `make LAB=1 EXTRA=-DG3_TIMELINE` turns them into stores to a per-translation-unit buffer that
// tools/gemm3_timeline.py / tools/gemm3_resid_timeline.py read through smtts_debug_read_timeline*.
#pragma once
#ifdef G3_TIMELINE
static __device__ unsigned long long g3_tl_buf[1024 * 160];
#ifdef G3_TL_NO_K      // no per-k-tile stamps: an s_memtime costs the one-wave-per-SIMD loop of the 64x64 tile as much as the k-tile itself
#define G3_STAMPK(i) do { } while (0)
#else
#define G3_STAMPK(i) do { if (tid == 0 && blockIdx.x < 1024 && (i) < 146) g3_tl_buf[blockIdx.x * 160 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)   // k-tiles 0 .. 35
#endif
#define G3_STAMP(i) do { if (tid == 0 && blockIdx.x < 1024 && (i) < 160) g3_tl_buf[blockIdx.x * 160 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define G3_STAMP_FENCE(v) asm volatile("s_nop 0" :: "v"(v))
#define G3_STAMPR(i) do { if (tid == 0 && blockIdx.x < 1024) g3_tl_buf[blockIdx.x * 160 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // constant 100 MHz: calibrates the shader clock
#define EPI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g3_tl_buf[blockIdx.x * 160 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// per translation unit: smtts_debug_read_timeline<suffix> / smtts_debug_clear_timeline<suffix>
#define G3_TIMELINE_EXPORTS(suffix)                                                                              \
    extern "C" int smtts_debug_read_timeline##suffix(unsigned long long* host, int n) {                          \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g3_tl_buf), (size_t)n * 8);                             \
    }                                                                                                            \
    extern "C" int smtts_debug_clear_timeline##suffix(void) {                                                    \
        void* p = nullptr;                                                                                       \
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g3_tl_buf)) != hipSuccess) return 1;                              \
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 1024 * 160);                                    \
    }
#else
#define G3_STAMPK(i) do { } while (0)
#define G3_STAMP(i) do { } while (0)
#define G3_STAMPR(i) do { } while (0)
#define G3_STAMP_FENCE(v) do { } while (0)
#define EPI_STAMP(i) do { } while (0)
#define G3_TIMELINE_EXPORTS(suffix)
#endif
