// conv2 / conv3 input-gradient kernels of the product (FP32 MFMA) with parts switched off (igemm.hpp BDR_IGEMM_ABL): where does the time go?
// Build (one binary per ablation): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc -DBDR_IGEMM_ABL=<n> tools/probes/dx_abl.hip -o tools/probes/dx_abl_<n>.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnn_layers.hpp"

using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef BDR_IGEMM_ABL
#define BDR_IGEMM_ABL 0
#endif

static float* dev_rand(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
template <class F>
static double time_us(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 100; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / 100;
}

int main()
{
    const int B = 256;
    const size_t n1 = (size_t)B * 400 * 32, n2 = (size_t)B * 81 * 64, n3 = (size_t)B * 49 * 64;
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    float* w3 = dev_rand(576 * 64, -0.05f, 0.05f, 4);
    float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
    float* dy3 = dev_rand(n3, -1.f, 1.f, 6);
    float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
    float* mask2 = dev_rand(n2, -1.f, 1.f, 7);
    float *dx1, *dx2; CK(hipMalloc(&dx1, n1 * 4)); CK(hipMalloc(&dx2, n2 * 4));
    DxArgs d2{dy2, w2, mask1, dx1, B * 100, nullptr, 0};
    DxArgs d3{dy3, w3, mask2, dx2, B * 81, nullptr, 0};
    const char* what[8] = {"full", "no k loop", "no stores", "no k loop, no stores", "no mask loads", "no k loop, no mask loads", "no stores, no mask loads", "launch + row maps only"};
    printf("ablation %d (%s)\n", BDR_IGEMM_ABL, what[BDR_IGEMM_ABL & 7]);
    printf("  conv2 dX merged classes, 64 x 128 tiles, 400 workgroups : %7.2f us\n", time_us([&] { CK((launch_igemm<DxC2M, 1>(0, dim3(m_tiles<DxC2M>(d2.M) * n_tiles<DxC2M>(), 1, 1), d2))); }));
    printf("  conv3 dX position classes, 2 teams                      : %7.2f us\n",
           time_us([&] { CK((launch_igemm<DxC3Pos, 2>(0, dim3(((B + DxC3Pos::WM * DxC3Pos::TM * 32 - 1) / (DxC3Pos::WM * DxC3Pos::TM * 32)) * n_tiles<DxC3Pos>(), 81, 1), d3))); }));
    // one workgroup per CU for the position-class kernel: extra dynamic LDS so that a second workgroup does not fit beside the first (the
    // 68 workgroups behind the first 256 then start where a SHORT workgroup has finished, not beside an interior position's 9 iterations)
    for (unsigned extra : {0u, 8u << 10, 16u << 10, 24u << 10}) {
        const dim3 g(((B + DxC3Pos::WM * DxC3Pos::TM * 32 - 1) / (DxC3Pos::WM * DxC3Pos::TM * 32)) * n_tiles<DxC3Pos>(), 81, 1);
        printf("  conv3 dX position classes, 2 teams, +%2u KB dynamic LDS   : %7.2f us\n", extra >> 10, time_us([&] {
            hipExtLaunchKernelGGL((k_igemm<DxC3Pos, 2>), g, dim3(64 * DxC3Pos::WM * DxC3Pos::WN * 2), extra, 0, nullptr, nullptr, 0, d3);
            CK(hipGetLastError()); }));
    }
    return 0;
}
