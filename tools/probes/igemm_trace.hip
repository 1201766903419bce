// Phase timeline of one k_igemm launch: when does each workgroup start, finish its prologue, its k loop,
// its epilogue?  (100 MHz wall clock, comparable across CUs.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIGEMM_TRACE -Iinclude -Iborder_amd/csrc -Itools/probes tools/probes/igemm_trace.hip -o tools/probes/igemm_trace.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnn_layers.hpp"
#include "igemm_abl.hpp"   // instrumented copy of k_igemm (namespace bdr_abl)
#include "igemm_dl.hpp"

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

template <class P, int TEAMS, bool DL = false>
static void trace(const char* name, dim3 grid, const typename P::Args& args)
{
    auto launch = [&]() { if constexpr (DL) return launch_igemm_dl<P, TEAMS>(0, grid, args); else return bdr_abl::launch_igemm<P, TEAMS>(0, grid, args); };
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* d; CK(hipMalloc(&d, nwg * 8 * 8)); CK(hipMemset(d, 0, nwg * 8 * 8));
    unsigned long long* null = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bdr_abl::g_igemm_trace), &null, sizeof(d)));
    for (int i = 0; i < 3; ++i) CK(launch());
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bdr_abl::g_igemm_trace), &d, sizeof(d)));
    CK(launch());
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8); CK(hipMemcpy(h.data(), d, nwg * 8 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    for (size_t i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 8]); tend = std::max(tend, h[i * 8 + 3]); }
    auto us = [&](unsigned long long t) { return (double)(t - t0) * 0.01; };
    std::vector<double> start(nwg), pro(nwg), loop(nwg), epi(nwg), end(nwg), mhz(nwg), cyc(nwg);
    for (size_t i = 0; i < nwg; ++i) {
        cyc[i] = (double)(h[i * 8 + 6] - h[i * 8 + 5]);
        mhz[i] = cyc[i] / ((h[i * 8 + 2] - h[i * 8 + 1]) * 0.01);
    }
    for (size_t i = 0; i < nwg; ++i) {
        start[i] = us(h[i * 8]); pro[i] = (h[i * 8 + 1] - h[i * 8]) * 0.01; loop[i] = (h[i * 8 + 2] - h[i * 8 + 1]) * 0.01;
        epi[i] = (h[i * 8 + 3] - h[i * 8 + 2]) * 0.01; end[i] = us(h[i * 8 + 3]);
    }
    auto stat = [&](std::vector<double> v, const char* nm) {
        std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= v.size();
        printf("   %-9s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f  mean %6.2f\n", nm, v.front(), v[v.size() / 10], v[v.size() / 2],
               v[v.size() * 9 / 10], v.back(), m);
    };
    printf("%s: %zu workgroups x %d threads, span %.2f us\n", name, nwg, 64 * P::WM * P::WN * TEAMS, us(tend));
    stat(start, "start"); stat(pro, "prologue"); stat(loop, "k-loop"); stat(epi, "epilogue"); stat(end, "end"); stat(cyc, "loop-cyc"); stat(mhz, "MHz");
    fflush(stdout);
    CK(hipFree(d));
}

int main(int argc, char** argv)
{
    const bool solo_only = argc > 1;
    printf("IGEMM_ABL=%d\n", IGEMM_ABL);
    const int B = 256, NZ = 2;
    float* x1 = dev_rand((size_t)B * 400 * 32, 0.f, 1.f, 1);
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    float* b2 = dev_rand(64, -0.1f, 0.1f, 3);
    float* h2[2]; for (int z = 0; z < 2; ++z) CK(hipMalloc(&h2[z], (size_t)B * 81 * 64 * 4));
    float* w3 = dev_rand(576 * 64, -0.05f, 0.05f, 4);
    float* h3[2]; for (int z = 0; z < 2; ++z) CK(hipMalloc(&h3[z], (size_t)B * 49 * 64 * 4));
    float* x2 = dev_rand((size_t)B * 81 * 64, 0.f, 1.f, 5);
    FwdArgs f2{}; for (int z = 0; z < NZ; ++z) { f2.x[z] = x1; f2.w[z] = w2; f2.bias[z] = b2; f2.out[z] = h2[z]; }
    f2.M = B * 81;
    FwdArgs f3{}; for (int z = 0; z < NZ; ++z) { f3.x[z] = x2; f3.w[z] = w3; f3.bias[z] = b2; f3.out[z] = h3[z]; }
    f3.M = B * 49;
    if (solo_only) {
        { FwdArgs g = f2; g.M = 64 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 3, true>("DL3 fwd_c2 64x64 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
        { FwdArgs g = f2; g.M = 64 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 4, true>("DL4 fwd_c2 64x64 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
        { FwdArgs g = f2; g.M = 128 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 2, 1>; trace<P, 3, true>("DL3 fwd_c2 128x64 tm2 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
        { using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 3, true>("DL3 fwd_c2 64x64 flat", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
        { using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 4, true>("DL4 fwd_c2 64x64 flat", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
        { FwdArgs g = f2; g.M = 64 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
        { FwdArgs g = f2; g.M = 128 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 2, 1>; trace<P, 1>("fwd_c2 128x64 tm2 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
        { FwdArgs g = f2; g.M = 64 * 256; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 duo (2 WG/CU)", dim3(256, 1, NZ), g); }
        { FwdArgs g = f2; g.M = 64 * 384; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 trio (3 WG/CU)", dim3(384, 1, NZ), g); }
        { using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 flat", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
        return 0;
    }
    { using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 flat", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
    { using P = FwdP<GeomC2, AFwd<GeomC2>, 3, 2, false, 96, 1, 1>; trace<P, 1>("fwd_c2 96x64 per-image", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
    { using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 2, 1>; trace<P, 1>("fwd_c2 128x64 tm2", dim3(m_tiles<P>(f2.M), 1, NZ), f2); }
    { using P = FwdP<GeomC3, AFwd<GeomC3>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c3 64x64 flat", dim3(m_tiles<P>(f3.M), 1, NZ), f3); }
    { using P = FwdP<GeomC3, AFwd<GeomC3>, 2, 2, false, 0, 1, 1>; trace<P, 2>("fwd_c3 64x64 flat teams2", dim3(m_tiles<P>(f3.M), 1, NZ), f3); }
    { FwdArgs g = f2; g.M = 64 * 128; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
    { FwdArgs g = f2; g.M = 64 * 256; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>; trace<P, 1>("fwd_c2 64x64 duo (2 WG/CU)", dim3(256, 1, NZ), g); }
    const size_t n2 = (size_t)B * 81 * 64, n1 = (size_t)B * 400 * 32;
    float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
    float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
    float* dx1; CK(hipMalloc(&dx1, n1 * 4));
    DxArgs d2{dy2, w2, mask1, dx1, B * 100};
    { using P = DxC2P<4, 1>; trace<P, 1>("dx_c2 128x32 flat", dim3(m_tiles<P>(d2.M), 4, 1), d2); }
    return 0;
}
