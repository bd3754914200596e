// Microbenchmark: do the matrix pipe (v_mfma) and the vector ALU overlap on one SIMD of MI355X, and does it matter whether
// the two waves that share a SIMD run their MFMA / VALU phases in phase or in anti-phase?
// One 512-thread workgroup per CU (8 waves = 2 per SIMD).  A "step" = NM dependent v_mfma_f32_32x32x16_bf16 (one accumulator)
// followed by NV dependent v_pk_fma_f32 + NT v_exp_f32 — the shape of one hidden tile of the fused codec FFN.
//   mode 0: MFMA phase only           mode 1: VALU phase only
//   mode 2: both phases, all waves start with MFMA (in phase)
//   mode 3: both phases, waves 4..7 start with the VALU phase (anti-phase with their SIMD partner)
//   mode 4: waves 0..3 only MFMA, waves 4..7 only VALU (each does 2x its share)
//   mode 5: one instruction stream interleaving the two (1 MFMA : NV/NM VALU), all waves
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NM = 24, NV = 192, NT = 32;

// VALU work runs as four independent dependency chains (GELU has eight value pairs in flight), MFMAs as one accumulator chain
__device__ __forceinline__ void mfma_phase(floatx16& acc, bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
struct V4 { f32x2 v[4]; float e[4]; };
// KIND of the NV vector instructions: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_and_b32, 3 v_cvt_pk_bf16_f32, 4 v_pk_mul_f32, 5 none (only the NT v_exp_f32),
// 6 v_pk_fma_f16, 7 v_pk_max_f16, 8 v_cvt_pk_f16_f32;  -DTEXP16: the NT transcendentals are v_exp_f16
#ifndef KIND
#define KIND 0
#endif
__device__ __forceinline__ void valu_one(V4& s, f32x2 c, int i) {
    if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(s.v[i & 3]) : "v"(c));
    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s.e[i & 3]) : "v"(c.x));
    if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(s.e[i & 3]) : "v"(c.x));
    if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(s.e[i & 3]) : "v"(c.x));
    if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(s.v[i & 3]) : "v"(c));
    if (KIND == 6) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(s.e[i & 3]) : "v"(c.x));   // round 4: packed fp16 (GELU in fp16?)
    if (KIND == 7) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(s.e[i & 3]) : "v"(c.x));
    if (KIND == 8) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(s.e[i & 3]) : "v"(c.x));
}
__device__ __forceinline__ void texp(float& v) {
#ifdef TEXP16
    asm volatile("v_exp_f16 %0, %0" : "+v"(v));
#else
    asm volatile("v_exp_f32 %0, %0" : "+v"(v));
#endif
}
__device__ __forceinline__ void valu_phase(V4& s, f32x2 c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) valu_one(s, c, i);
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(s.e[i & 3]));
}

__global__ __launch_bounds__(512) void k(int steps, int mode, float* sink, unsigned long long* cyc, unsigned* hwid) {
    if (mode >= 10) mode -= 10;  // launched with 256 threads: one wave per SIMD
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    V4 st;
    for (int i = 0; i < 4; ++i) { st.v[i].x = lane * 1e-3f + i; st.v[i].y = 0.5f; st.e[i] = 0.25f * i; }
    f32x2& v = st.v[0];
    float& ex = st.e[0];
    f32x2 c; c.x = 0.999f; c.y = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    const bool second = wave >= 4;
    if (mode == 0) {
        for (int s = 0; s < steps; ++s) { mfma_phase(acc, a, b); __builtin_amdgcn_sched_barrier(0); }
    } else if (mode == 1) {
        for (int s = 0; s < steps; ++s) { valu_phase(st, c); __builtin_amdgcn_sched_barrier(0); }
    } else if (mode == 2 || (mode == 3 && !second)) {
        for (int s = 0; s < steps; ++s) {
            mfma_phase(acc, a, b);
            __builtin_amdgcn_sched_barrier(0);
            v.x += acc[0];  // the VALU phase waits for the MFMA result, as GELU does
            valu_phase(st, c);
            __builtin_amdgcn_sched_barrier(0);
            a[0] = (__bf16)v.y;  // and the next MFMA phase for the VALU result
        }
    } else if (mode == 3) {
        for (int s = 0; s < steps; ++s) {
            valu_phase(st, c);
            __builtin_amdgcn_sched_barrier(0);
            a[0] = (__bf16)v.y;
            mfma_phase(acc, a, b);
            __builtin_amdgcn_sched_barrier(0);
            v.x += acc[0];
        }
    } else if (mode == 4) {
        if (!second) for (int s = 0; s < 2 * steps; ++s) { mfma_phase(acc, a, b); __builtin_amdgcn_sched_barrier(0); }
        else for (int s = 0; s < 2 * steps; ++s) { valu_phase(st, c); __builtin_amdgcn_sched_barrier(0); }
    } else if (mode == 5) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) valu_one(st, c, j);
                if (i < NT) texp(st.e[i & 3]);
                if (i + NM < NT) texp(st.e[(i + 1) & 3]);
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    const unsigned long long c1 = clock64();
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = c1 - c0; hwid[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)); }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) sum += st.v[i].x + st.v[i].y + st.e[i];
    for (int r = 0; r < 16; ++r) sum += acc[r];
    if (sum == 123.456f) *sink = sum;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    const int lds = argc > 2 ? atoi(argv[2]) : 96 * 1024;  // > 80 KiB: one workgroup per CU (otherwise the dispatcher may stack two on one CU)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&sink, 4);
    unsigned* hwid;
    hipMalloc(&hwid, 256 * 8 * 4);
    hipMalloc(&cyc, 256 * 8 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"MFMA only", "VALU only", "both, in phase", "both, anti-phase", "4 MFMA waves + 4 VALU waves (2x each)", "one interleaved stream"};
    printf("KIND %d: ", KIND);
    printf("step = %d MFMA 32x32x16 (%d matrix clks) + %d v_pk_fma_f32 + %d v_exp_f32; 8 waves / CU, 256 CUs, %d steps\n", NM, NM * 32, NV, NT, steps);
    const int modes[] = {4, 10, 11, 12, 15};
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(mode >= 10 ? 256 : 512), lds, 0, steps, mode, sink, cyc, hwid);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long hh[256 * 8 * 2];
        hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
        unsigned long long h[2] = {hh[0], hh[1]}, wmax = 0, wmin = ~0ull;
        for (int i = 0; i < 256 * (mode >= 10 ? 4 : 8); ++i) {
            const int idx = (i / (mode >= 10 ? 4 : 8)) * 8 + i % (mode >= 10 ? 4 : 8);
            wmax = hh[2 * idx] > wmax ? hh[2 * idx] : wmax;
            wmin = hh[2 * idx] < wmin ? hh[2 * idx] : wmin;
        }
        static unsigned hw[256 * 8];
        hipMemcpy(hw, hwid, sizeof(hw), hipMemcpyDeviceToHost);
        int hist[9] = {0};  // workgroups by the largest number of their waves that share one SIMD
        for (int b = 0; b < 256; ++b) {
            int per[4] = {0, 0, 0, 0};
            for (int w = 0; w < (mode >= 10 ? 4 : 8); ++w) per[(hw[b * 8 + w] >> 4) & 3]++;
            int mx = 0;
            for (int i = 0; i < 4; ++i) mx = per[i] > mx ? per[i] : mx;
            hist[mx]++;
        }
        printf("   SIMD of waves 0..7 of workgroup 0:");
        for (int w = 0; w < (mode >= 10 ? 4 : 8); ++w) printf(" %u", (hw[w] >> 4) & 3);
        printf(" | workgroups by max waves on one SIMD: 1:%d 2:%d 3:%d 4:%d 5+:%d\n", hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7] + hist[8]);
        printf("   wave wall time min %.1f max %.1f us | ", wmin * 0.01, wmax * 0.01);
        printf("mode %2d  %-42s %s %8.1f us  = %7.0f ns per step = %6.0f shader clks (%.2f GHz)\n", mode, names[mode % 10],
               mode >= 10 ? "1 wave/SIMD " : "2 waves/SIMD", ms * 1e3, ms * 1e6 / steps, (double)h[1] / steps, h[1] / (h[0] * 10.0));
    }
    return 0;
}
