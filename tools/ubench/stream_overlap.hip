// Do kernels from different HIP streams run side by side?  A kernel of G workgroups that each spin for ~T us, launched
// N times round-robin on S streams: with real concurrency the time per launch drops ~S-fold while S * G workgroups fit.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_overlap.hip -o /tmp/stream_overlap && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int* sink) {  // wall_clock64: 100 MHz
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks == 0xdeadbeefULL) *sink = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipStream_t st[4];
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 300;
    for (int G : {8, 128, 512}) for (int lds : {0, 65536}) for (int S : {1, 2, 3, 4}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(G), dim3(256), lds, st[i % S], 5000ULL, sink);  // 50 us
            hipDeviceSynchronize();
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("G=%4d workgroups x 256 thr, LDS %5d B, 50 us each: %d stream(s) -> %6.1f us per launch\n", G, lds, S, us);
        }
    }
    return 0;
}
