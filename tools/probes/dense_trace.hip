// Phase timeline of the dense (SAC-sized) k_igemm launches: where do the ~8 us of a 1024 x 256 x 256 layer go?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIGEMM_TRACE -Iinclude -Iborder_amd/csrc tools/probes/dense_trace.hip -o tools/probes/dense_trace.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dense.hpp"

using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <class P, int TEAMS>
static void trace(const char* name, dim3 grid, const typename P::Args& args)
{
    auto launch = [&]() { return launch_igemm<P, TEAMS>(0, grid, args); };
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* d; CK(hipMalloc(&d, nwg * 8 * 8)); CK(hipMemset(d, 0, nwg * 8 * 8));
    unsigned long long* null = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_trace), &null, sizeof(d)));
    for (int i = 0; i < 3; ++i) CK(launch());
    CK(hipDeviceSynchronize());
    // back-to-back launch time without tracing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) CK(launch());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_trace), &d, sizeof(d)));
    CK(launch());
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8); CK(hipMemcpy(h.data(), d, nwg * 8 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    for (size_t i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 8]); tend = std::max(tend, h[i * 8 + 3]); }
    auto us = [&](unsigned long long t) { return (double)(t - t0) * 0.01; };
    std::vector<double> start(nwg), pro(nwg), loop(nwg), epi(nwg), end(nwg), mhz(nwg);
    for (size_t i = 0; i < nwg; ++i) {
        mhz[i] = (double)(h[i * 8 + 6] - h[i * 8 + 5]) / ((h[i * 8 + 2] - h[i * 8 + 1]) * 0.01);
        start[i] = us(h[i * 8]); pro[i] = (h[i * 8 + 1] - h[i * 8]) * 0.01; loop[i] = (h[i * 8 + 2] - h[i * 8 + 1]) * 0.01;
        epi[i] = (h[i * 8 + 3] - h[i * 8 + 2]) * 0.01; end[i] = us(h[i * 8 + 3]);
    }
    auto stat = [&](std::vector<double> v, const char* nm) {
        std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= v.size();
        printf("   %-9s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f  mean %6.2f\n", nm, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back(), m);
    };
    printf("%s: %zu workgroups x %d threads, traced span %.2f us, back-to-back %.2f us/launch\n", name, nwg, 64 * P::WM * P::WN * TEAMS, us(tend), ms * 1000 / 200);
    stat(start, "start"); stat(pro, "prologue"); stat(loop, "k-loop"); stat(epi, "epilogue"); stat(end, "end"); stat(mhz, "MHz");
    fflush(stdout);
    CK(hipFree(d));
}

int main()
{
    const int M = 1024;
    for (int K : {64, 256}) for (int N : {64, 256}) {
        float* x = dev_rand((size_t)M * K, -1.f, 1.f, 1);
        float* w = dev_rand((size_t)K * N + N, -0.05f, 0.05f, 2);
        float* out[4]; for (int z = 0; z < 4; ++z) CK(hipMalloc(&out[z], (size_t)M * N * 4));
        auto mk = [&](int z) { DenseArgs d{}; d.x = DenseSrc{x, K}; d.w = w; d.bias = w + (size_t)K * N; d.out = out[z]; d.ldo = N; d.M = M; d.ncols = N; d.kred = K; d.relu = 1; d.w_ld = N; d.had_group = 1; return d; };
        char nm[128];
        const dim3 g1((M / 64) * (N / 64), 1, 1);
        snprintf(nm, sizeof nm, "fwd %dx%dx%d z1 teams1", M, K, N); trace<DenseFwd, 1>(nm, g1, mk(0));
        snprintf(nm, sizeof nm, "fwd %dx%dx%d z1 teams2", M, K, N); trace<DenseFwd, 2>(nm, g1, mk(0));
        DenseArgsZ dz{}; for (int z = 0; z < 4; ++z) dz.a[z] = mk(z);
        const dim3 g4((M / 64) * (N / 64), 1, 4);
        snprintf(nm, sizeof nm, "fwd %dx%dx%d z4 teams1", M, K, N); trace<DenseFwdZ, 1>(nm, g4, dz);
        snprintf(nm, sizeof nm, "fwd %dx%dx%d z4 teams2", M, K, N); trace<DenseFwdZ, 2>(nm, g4, dz);
    }
    return 0;
}
