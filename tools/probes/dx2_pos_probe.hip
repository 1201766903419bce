// conv2 input gradient (round 6): the merged-class GEMM over flat row tiles (DxC2M, rounds 4-5) against position-class tiles (DxC2MPos: only
// the taps that reach a valid output) - bitwise comparison of the results and time per launch, B = 256 and a ragged batch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc tools/probes/dx2_pos_probe.hip -o tools/probes/dx2_pos_probe.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cnn_layers.hpp"

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
template <class F>
static double time_us(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / 200;
}
template <class P, int T>
static void run_pos(const char* name, int B, const DxArgs& d, float* out, const std::vector<float>& ref)
{
    const size_t n1 = (size_t)B * 400 * 32;
    CK(hipMemset(out, 0xff, n1 * 4));
    CK((launch_igemm<P, T>(0, dxc2_pos_grid<P>(B), d)));
    std::vector<float> got(n1); CK(hipMemcpy(got.data(), out, n1 * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < n1; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
    const dim3 g = dxc2_pos_grid<P>(B);
    printf("  %-46s grid (%u,%u) x %d thr: %7.2f us   %zu of %zu words differ from the flat-tile kernel\n", name, g.x, g.y, 64 * P::WM * P::WN * T,
           time_us([&] { CK((launch_igemm<P, T>(0, g, d))); }), bad, n1);
    fflush(stdout);
}
int main()
{
    for (int B : {256, 77}) {
        const size_t n1 = (size_t)B * 400 * 32, n2 = (size_t)B * 81 * 64;
        float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
        float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
        float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
        float *ref_d, *out; CK(hipMalloc(&ref_d, n1 * 4)); CK(hipMalloc(&out, n1 * 4));
        DxArgs d{dy2, w2, mask1, ref_d, B * 100, nullptr, 0};
        CK((launch_igemm<DxC2M, 1>(0, dim3(m_tiles<DxC2M>(d.M) * n_tiles<DxC2M>(), 1, 1), d)));
        std::vector<float> ref(n1); CK(hipMemcpy(ref.data(), ref_d, n1 * 4, hipMemcpyDeviceToHost));
        printf("B = %d\n  %-46s grid (%d) x 256 thr: %7.2f us\n", B, "flat row tiles 64 x 128 (DxC2M, rounds 4-5)", m_tiles<DxC2M>(d.M) * n_tiles<DxC2M>(),
               time_us([&] { CK((launch_igemm<DxC2M, 1>(0, dim3(m_tiles<DxC2M>(d.M) * n_tiles<DxC2M>(), 1, 1), d))); }));
        d.out = out;
        run_pos<DxC2MPosP<2, 2, 1, 2>, 1>("position tiles 64 img x 128 (2,2,1,2)", B, d, out, ref);
        run_pos<DxC2MPosP<2, 2, 1, 2>, 2>("position tiles 64 img x 128, 2 teams", B, d, out, ref);
        run_pos<DxC2MPosP<1, 4, 1, 1>, 1>("position tiles 32 img x 128 (1,4,1,1)", B, d, out, ref);
        run_pos<DxC2MPosP<1, 2, 1, 2>, 1>("position tiles 32 img x 128 (1,2,1,2) 2 waves", B, d, out, ref);
        run_pos<DxC2MPosP<2, 2, 1, 1>, 1>("position tiles 64 img x 64 (2,2,1,1)", B, d, out, ref);
        run_pos<DxC2MPosP<2, 1, 1, 2>, 1>("position tiles 64 img x 64 (2,1,1,2) 2 waves", B, d, out, ref);
        run_pos<DxC2MPosP<4, 1, 1, 2>, 1>("position tiles 128 img x 64 (4,1,1,2)", B, d, out, ref);
        run_pos<DxC2MPosP<2, 2, 2, 2>, 1>("position tiles 128 img x 128 (2,2,2,2)", B, d, out, ref);
    }
    return 0;
}
