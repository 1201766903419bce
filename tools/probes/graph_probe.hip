// How much does a hipGraph buy for a chain of small dependent kernels (the SAC / CartPole step shape)?
// eager launches vs one graph launch per step, N kernels per step, grids of 1 / 64 / 256 workgroups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_small(float* p, int n, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
int main()
{
    float* d; CK(hipMalloc(&d, 1 << 22));
    CK(hipMemset(d, 0, 1 << 22));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int N : {16, 72}) for (int wgs : {1, 64, 256}) {
        const int steps = 400;
        auto run_eager = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, st, d, wgs * 256, 64); };
        for (int w = 0; w < 20; ++w) run_eager();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) run_eager();
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        auto t2 = std::chrono::steady_clock::now();
        const double host_e = std::chrono::duration<double, std::micro>(t1 - t0).count() / steps, tot_e = std::chrono::duration<double, std::micro>(t2 - t0).count() / steps;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        run_eager();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) CK(hipGraphLaunch(ge, st));
        t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        t2 = std::chrono::steady_clock::now();
        const double host_g = std::chrono::duration<double, std::micro>(t1 - t0).count() / steps, tot_g = std::chrono::duration<double, std::micro>(t2 - t0).count() / steps;
        printf("N=%2d wgs=%3d  eager: host %.1f us/step total %.1f us/step (%.2f us/kernel) | graph: host %.1f total %.1f (%.2f us/kernel)\n", N, wgs, host_e, tot_e,
               tot_e / N, host_g, tot_g, tot_g / N);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
