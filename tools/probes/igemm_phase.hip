// Where do the cycles of one wave go inside the k loop of k_igemm?  (s_memtime phase sums, conv2 forward, B = 256)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIGEMM_TRACE2 -Iinclude -Iborder_amd/csrc -Itools/probes tools/probes/igemm_phase.hip -o tools/probes/igemm_phase.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cnn_layers.hpp"
#include "igemm_abl.hpp"   // instrumented copy of k_igemm (namespace bdr_abl)
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
static void phases(const char* name, dim3 grid, const typename P::Args& args)
{
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* d; CK(hipMalloc(&d, nwg * 64)); CK(hipMemset(d, 0, nwg * 64));
    for (int i = 0; i < 3; ++i) CK((bdr_abl::launch_igemm<P, TEAMS>(0, grid, args)));
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bdr_abl::g_igemm_phase), &d, sizeof(d)));
    CK((bdr_abl::launch_igemm<P, TEAMS>(0, grid, args)));
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8); CK(hipMemcpy(h.data(), d, nwg * 64, hipMemcpyDeviceToHost));
    double s[5] = {0, 0, 0, 0, 0}, n = 0;
    for (size_t i = 0; i < nwg; ++i) { for (int q = 0; q < 5; ++q) s[q] += (double)h[i * 8 + q]; n += (double)h[i * 8 + 5]; }
    printf("%-30s %zu WGs, cycles per k-tile (wave 0):  frag-wait+4 MFMA %6.0f | commit %6.0f | 12 MFMA+prefetch %6.0f | tail %5.0f | barrier %6.0f | total %6.0f\n",
           name, nwg, s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n, (s[0] + s[1] + s[2] + s[3] + s[4]) / n);
    unsigned long long* null = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(bdr_abl::g_igemm_phase), &null, sizeof(null)));
    CK(hipFree(d));
}
int main()
{
    const int B = 256, NZ = 2;
    float* x1 = dev_rand((size_t)B * 400 * 32, 0.f, 1.f, 1);
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    float* b2 = dev_rand(64, -0.1f, 0.1f, 3);
    float* h2[2]; for (int z = 0; z < 2; ++z) CK(hipMalloc(&h2[z], (size_t)B * 81 * 64 * 4));
    FwdArgs f2{}; for (int z = 0; z < NZ; ++z) { f2.x[z] = x1; f2.w[z] = w2; f2.bias[z] = b2; f2.out[z] = h2[z]; }
    using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false>;
    { FwdArgs g = f2; g.M = 64 * 128; phases<P, 1>("fwd_c2 solo (1 WG/CU)", dim3(128, 1, NZ), g); }
    { FwdArgs g = f2; g.M = 64 * 256; phases<P, 1>("fwd_c2 duo (2 WG/CU)", dim3(256, 1, NZ), g); }
    f2.M = B * 81;
    phases<P, 1>("fwd_c2 real (648 WGs)", dim3((f2.M + 63) / 64, 1, NZ), f2);
    return 0;
}
