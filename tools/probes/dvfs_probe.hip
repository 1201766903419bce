// How does the shader clock behave when work arrives after an idle period?  A chip-filling FP32-MFMA kernel (~60 us) is launched
// back to back after sleeping; every launch records wall clock (100 MHz s_memrealtime) and shader clock (s_memtime) deltas.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dvfs_probe.hip -o tools/probes/dvfs_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_burn(unsigned long long* out, int slot, int iters, float* sink)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) sink[0] = acc[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[slot * 2] = wall_clock64() - w0; out[slot * 2 + 1] = clock64() - c0; }
}
int main()
{
    const int N = 400;
    unsigned long long* d; hipMalloc(&d, N * 16); float* sink; hipMalloc(&sink, 64);
    hipStream_t st; hipStreamCreate(&st);
    for (int idle_ms : {0, 1, 10, 100, 1000}) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_burn, dim3(1024), dim3(256), 0, st, d, 0, 1500, sink);   // get hot
        hipStreamSynchronize(st);
        std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_burn, dim3(1024), dim3(256), 0, st, d, i, 1500, sink);
        hipStreamSynchronize(st);
        std::vector<unsigned long long> h(N * 2); hipMemcpy(h.data(), d, N * 16, hipMemcpyDeviceToHost);
        printf("idle %4d ms: kernel us / MHz at launch", idle_ms);
        double t = 0;
        for (int i = 0; i < N; ++i) {
            const double us = h[i * 2] / 100.0, mhz = h[i * 2 + 1] / us;
            if (i < 4 || i == 8 || i == 16 || i == 32 || i == 64 || i == 128 || i == 256 || i == N - 1) printf("  #%d(t=%.1fms) %.1f/%.0f", i, t / 1e3, us, mhz);
            t += us;
        }
        printf("\n");
    }
    return 0;
}
