// What does the shader clock do under a LAUNCH-BOUND load (a chain of small dependent kernels, SAC-like) compared with a chip-filling
// one?  Every kernel records wall clock (100 MHz s_memrealtime) and shader clock (s_memtime) over its own lifetime; the ratio is the
// clock the kernel ran at.  (not part of the product)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/light_load_clock.hip -o tools/probes/light_load_clock.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_work(unsigned long long* out, int slot, int iters, float* sink)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) sink[0] = acc[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[slot * 2] = wall_clock64() - w0; out[slot * 2 + 1] = clock64() - c0; }
}
static void run(const char* name, hipStream_t st, unsigned long long* d, float* sink, int grid, int iters, int n)
{
    for (int rep = 0; rep < 3; ++rep) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, st, d, i, iters, sink);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, st, d, i, iters, sink);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(n * 2); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    std::vector<double> us, mhz;
    for (int i = n / 2; i < n; ++i) { us.push_back(h[i * 2] / 100.0); mhz.push_back(h[i * 2 + 1] / (h[i * 2] / 100.0)); }
    std::sort(us.begin(), us.end()); std::sort(mhz.begin(), mhz.end());
    printf("%-44s grid %5d x %4d MFMAs/wave: %.2f us per launch (stream), in-kernel %.2f us, shader clock %.0f MHz (median; min %.0f max %.0f)\n", name, grid, iters,
           1000.0 * ms / n, us[us.size() / 2], mhz[mhz.size() / 2], mhz.front(), mhz.back());
}
int main()
{
    const int N = 4000;
    unsigned long long* d; hipMalloc(&d, N * 16); float* sink; hipMalloc(&sink, 64);
    hipStream_t st; hipStreamCreate(&st);
    run("chip-filling, long (4096 WG, 1500 MFMA)", st, d, sink, 4096, 1500, 400);
    run("SAC-like: small grid, short (256 WG, 32 MFMA)", st, d, sink, 256, 32, N);
    run("SAC-like: 1024 WG, 32 MFMA", st, d, sink, 1024, 32, N);
    run("one workgroup, 32 MFMA (C1-like)", st, d, sink, 1, 32, N);
    run("one workgroup, 2000 MFMA (C1-like, long)", st, d, sink, 1, 2000, N);
    run("chip-filling again", st, d, sink, 4096, 1500, 400);
    return 0;
}
