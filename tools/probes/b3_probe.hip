// 3xbf16 split-operand implicit GEMM (igemm_b3.hpp) vs the FP32-MFMA kernel: accuracy and time (B=256 Atari shapes).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc tools/probes/b3_probe.hip -o tools/probes/b3_probe.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnn_layers.hpp"
#include "igemm_b3.hpp"

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

struct FwdB3Args : FwdArgs { const uint16_t* wpl[MAXZ]; };
template <class G, int WM_, int WN_, int TM_ = 1, int TN_ = 1>
struct FwdB3 : FwdP<G, AFwd<G>, WM_, WN_, false, 0, TM_, TN_> {
    using Args = FwdB3Args;
    __device__ static const uint4* b_chunk(const Args& a, int z, int, int pl, int kt, int n, int kq)
    {
        return reinterpret_cast<const uint4*>(a.wpl[z] + (size_t)pl * G::COUT * G::K + (size_t)n * G::K + kt * 32 + kq * 8);
    }
};
struct DxB3Args : DxArgs { const uint16_t* wpl; };
template <int WM_, int WN_>
struct DxC3B3 : DxC3P<WM_, WN_> {
    using Args = DxB3Args;
    using G = GeomC3;
    __device__ static const uint4* b_chunk(const Args& a, int, int y, int pl, int kt, int n, int kq)
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        return reinterpret_cast<const uint4*>(a.wpl + (size_t)pl * G::K * G::COUT + ((size_t)(tap * G::CIN + n) * G::COUT + c0 + kq * 8));
    }
};

template <int WM_, int WN_, int TM_, int TN_>
struct DxC2MB3 : DxC2MP<WM_, WN_, TM_, TN_> {
    using Base = DxC2MP<WM_, WN_, TM_, TN_>;
    using Args = DxB3Args;
    using G = GeomC2;
    __device__ static const uint4* b_chunk(const Args& a, int, int y, int pl, int kt, int n, int kq)
    {
        constexpr int TPT = G::COUT / BK;
        const int tap = kt / TPT, c0 = (kt % TPT) * BK;
        return reinterpret_cast<const uint4*>(a.wpl + (size_t)pl * G::K * G::COUT + ((size_t)Base::b_row(y, tap, n) * G::COUT + c0 + kq * 8));
    }
};

template <class F>
static double time_us(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / 50;
}
static void compare(const char* name, const float* a, const float* b, size_t n)
{
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)std::fabs(ha[i] - hb[i])); mx = std::max(mx, (double)std::fabs(hb[i])); }
    printf("   %-28s max |diff| %.3e, max |ref| %.3e, rel %.2e\n", name, md, mx, md / mx);
}

int main()
{
    const int B = 256, NZ = 2;
    float* x1 = dev_rand((size_t)B * 400 * 32, 0.f, 1.f, 1);
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    float* b2 = dev_rand(64, -0.1f, 0.1f, 3);
    float* w3 = dev_rand(576 * 64, -0.05f, 0.05f, 4);
    float* x2 = dev_rand((size_t)B * 81 * 64, 0.f, 1.f, 5);
    const size_t n2 = (size_t)B * 81 * 64, n3 = (size_t)B * 49 * 64;
    float *h2[2], *h2b[2], *h3[2], *h3b[2];
    for (int z = 0; z < 2; ++z) { CK(hipMalloc(&h2[z], n2 * 4)); CK(hipMalloc(&h2b[z], n2 * 4)); CK(hipMalloc(&h3[z], n3 * 4)); CK(hipMalloc(&h3b[z], n3 * 4)); }
    uint16_t *w2t, *w3t, *w3p;
    CK(hipMalloc(&w2t, 3 * 512 * 64 * 2)); CK(hipMalloc(&w3t, 3 * 576 * 64 * 2)); CK(hipMalloc(&w3p, 3 * 576 * 64 * 2));
    hipLaunchKernelGGL(k_split_planes, dim3((512 * 64 + 255) / 256), dim3(256), 0, 0, w2, w2t, 512, 64, 1);
    hipLaunchKernelGGL(k_split_planes, dim3((576 * 64 + 255) / 256), dim3(256), 0, 0, w3, w3t, 576, 64, 1);
    hipLaunchKernelGGL(k_split_planes, dim3((576 * 64 + 255) / 256), dim3(256), 0, 0, w3, w3p, 576, 64, 0);
    CK(hipDeviceSynchronize());

    FwdB3Args f2{}; f2.M = B * 81;
    for (int z = 0; z < NZ; ++z) { f2.x[z] = x1; f2.w[z] = w2; f2.bias[z] = b2; f2.out[z] = h2[z]; f2.wpl[z] = w2t; }
    FwdB3Args f2b = f2; for (int z = 0; z < NZ; ++z) f2b.out[z] = h2b[z];
    FwdB3Args f3{}; f3.M = B * 49;
    for (int z = 0; z < NZ; ++z) { f3.x[z] = x2; f3.w[z] = w3; f3.bias[z] = b2; f3.out[z] = h3[z]; f3.wpl[z] = w3t; }
    FwdB3Args f3b = f3; for (int z = 0; z < NZ; ++z) f3b.out[z] = h3b[z];

    {
        using PF = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false>;
        const dim3 g((f2.M + 63) / 64, 1, NZ);
        printf("fwd_c2  f32 MFMA          : %7.2f us\n", time_us([&] { CK((launch_igemm<PF, 1>(0, g, (const FwdArgs&)f2))); }));
        using P9 = FwdB3<GeomC2, 2, 2>;
        printf("fwd_c2  3xbf16, 9 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 9>(0, g, f2b))); }));
        compare("9 terms vs f32", h2b[1], h2[1], n2);
        printf("fwd_c2  3xbf16, 6 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 6>(0, g, f2b))); }));
        compare("6 terms vs f32", h2b[1], h2[1], n2);
        using PT = FwdB3<GeomC2, 2, 1, 1, 2>;   // 2 waves x (32 x 64)
        printf("fwd_c2  3xbf16 9t, 2x(32x64): %7.2f us\n", time_us([&] { CK((launch_igemm_b3<PT, 9>(0, g, f2b))); }));
        compare("9 terms tn2 vs f32", h2b[1], h2[1], n2);
        using PW = FwdB3<GeomC2, 4, 1, 1, 2>;   // 4 waves x (32 x 64): 128 x 64 tile
        const dim3 gw((f2.M + 127) / 128, 1, NZ);
        printf("fwd_c2  3xbf16 9t, 4x(32x64): %7.2f us\n", time_us([&] { CK((launch_igemm_b3<PW, 9>(0, gw, f2b))); }));
        compare("9 terms 128x64 vs f32", h2b[1], h2[1], n2);
    }
    {
        using PF = FwdP<GeomC3, AFwd<GeomC3>, 2, 2, false>;
        const dim3 g((f3.M + 63) / 64, 1, NZ);
        printf("fwd_c3  f32 MFMA          : %7.2f us\n", time_us([&] { CK((launch_igemm<PF, 1>(0, g, (const FwdArgs&)f3))); }));
        using P9 = FwdB3<GeomC3, 2, 2>;
        printf("fwd_c3  3xbf16, 9 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 9>(0, g, f3b))); }));
        compare("9 terms vs f32", h3b[1], h3[1], n3);
        printf("fwd_c3  3xbf16, 6 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 6>(0, g, f3b))); }));
        compare("6 terms vs f32", h3b[1], h3[1], n3);
    }
    {
        float* dy3 = dev_rand(n3, -1.f, 1.f, 6);
        float* mask2 = dev_rand(n2, -1.f, 1.f, 7);
        float *dx2, *dx2b; CK(hipMalloc(&dx2, n2 * 4)); CK(hipMalloc(&dx2b, n2 * 4));
        DxB3Args d3{}; d3.dy = dy3; d3.w = w3; d3.mask = mask2; d3.out = dx2; d3.M = B * 81; d3.wpl = w3p;
        DxB3Args d3b = d3; d3b.out = dx2b;
        const dim3 g((d3.M + 63) / 64, 1, 1);
        printf("dx_c3   f32 MFMA (2 teams): %7.2f us\n", time_us([&] { CK((launch_igemm<DxC3P<2, 2>, 2>(0, g, (const DxArgs&)d3))); }));
        using P9 = DxC3B3<2, 2>;
        printf("dx_c3   3xbf16, 9 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 9>(0, g, d3b))); }));
        compare("9 terms vs f32", dx2b, dx2, n2);
        printf("dx_c3   3xbf16, 6 terms   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P9, 6>(0, g, d3b))); }));
        compare("6 terms vs f32", dx2b, dx2, n2);
    }
    {   // conv2 input gradient over merged parity classes: [B*100][256] x [256][128]
        float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
        const size_t n1 = (size_t)B * 400 * 32;
        float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
        float *dx1, *dx1b; CK(hipMalloc(&dx1, n1 * 4)); CK(hipMalloc(&dx1b, n1 * 4));
        uint16_t* w2p; CK(hipMalloc(&w2p, 3 * 512 * 64 * 2));
        hipLaunchKernelGGL(k_split_planes, dim3((512 * 64 + 255) / 256), dim3(256), 0, 0, w2, w2p, 512, 64, 0);
        DxB3Args d2{}; d2.dy = dy2; d2.w = w2; d2.mask = mask1; d2.out = dx1; d2.M = B * 100; d2.wpl = w2p;
        DxB3Args d2b = d2; d2b.out = dx1b;
        using PF = DxC2MP<2, 2, 1, 2>;
        printf("dx_c2m  f32 MFMA 64x128   : %7.2f us\n", time_us([&] { CK((launch_igemm<PF, 1>(0, dim3(m_tiles<PF>(d2.M) * n_tiles<PF>(), 1, 1), (const DxArgs&)d2))); }));
        using P1 = DxC2MB3<2, 2, 1, 2>;
        printf("dx_c2m  3xbf16 6t 64x128  : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P1, 6>(0, dim3(m_tiles<P1>(d2.M) * n_tiles<P1>(), 1, 1), d2b))); }));
        compare("6 terms vs f32", dx1b, dx1, n1);
        using P2 = DxC2MB3<2, 2, 1, 1>;
        printf("dx_c2m  3xbf16 6t 64x64   : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P2, 6>(0, dim3(m_tiles<P2>(d2.M) * n_tiles<P2>(), 1, 1), d2b))); }));
        compare("6 terms vs f32", dx1b, dx1, n1);
        using P3 = DxC2MB3<4, 1, 1, 1>;
        printf("dx_c2m  3xbf16 6t 128x32  : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P3, 6>(0, dim3(m_tiles<P3>(d2.M) * n_tiles<P3>(), 1, 1), d2b))); }));
        compare("6 terms vs f32", dx1b, dx1, n1);
        using P4 = DxC2MB3<2, 2, 2, 1>;
        printf("dx_c2m  3xbf16 6t 128x64  : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P4, 6>(0, dim3(m_tiles<P4>(d2.M) * n_tiles<P4>(), 1, 1), d2b))); }));
        compare("6 terms vs f32", dx1b, dx1, n1);
        using P5 = DxC2MB3<1, 4, 1, 1>;
        printf("dx_c2m  3xbf16 6t 32x128  : %7.2f us\n", time_us([&] { CK((launch_igemm_b3<P5, 6>(0, dim3(m_tiles<P5>(d2.M) * n_tiles<P5>(), 1, 1), d2b))); }));
        compare("6 terms vs f32", dx1b, dx1, n1);
    }
    return 0;
}
