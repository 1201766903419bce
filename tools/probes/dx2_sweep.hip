// Tile-shape sweep of the conv2 input-gradient GEMM (4 parity classes, 25 600 rows x 32 x 256 each) with the plain k_igemm.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc tools/probes/dx2_sweep.hip -o tools/probes/dx2_sweep.bin
#include <cstdio>
#include <cstdlib>
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
static double checksum(const float* d, size_t n)
{
    std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    double s = 0; for (size_t i = 0; i < n; ++i) s += (double)h[i] * (double)((i % 97) + 1);
    return s;
}
template <class P, int TEAMS>
static void run(const char* name, dim3 grid, const typename P::Args& args, const float* out, size_t nout)
{
    hipStream_t st = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK((launch_igemm<P, TEAMS>(st, grid, args)));
    CK(hipDeviceSynchronize());
    const int IT = 200;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < IT; ++i) CK((launch_igemm<P, TEAMS>(st, grid, args)));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s grid=(%4u,%u,%u) thr=%4d  %7.2f us   sum=%.9e\n", name, grid.x, grid.y, grid.z, 64 * P::WM * P::WN * TEAMS, ms * 1000.0 / IT, checksum(out, nout));
    fflush(stdout);
}
int main()
{
    const int B = 256;
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    const size_t n2 = (size_t)B * 81 * 64, n1 = (size_t)B * 400 * 32;
    float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
    float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
    float* dx1; CK(hipMalloc(&dx1, n1 * 4));
    DxArgs d2{dy2, w2, mask1, dx1, B * 100};
#define DX2(WM, WN, RP, TM, TN, T) { using P = DxC2P<WM, WN, RP, TM, TN>; \
    CK(hipMemset(dx1, 0, n1 * 4)); run<P, T>("dx_c2 w" #WM "x" #WN " rp" #RP " t" #TM "x" #TN " teams" #T, dim3(m_tiles<P>(d2.M) * (32 / (WN * TN * 32)), 4, 1), d2, dx1, n1); }
    DX2(4, 1, 0, 1, 1, 1)   // current: 128x32 tiles
    DX2(4, 1, 0, 1, 1, 2)
    DX2(4, 1, 0, 2, 1, 1)   // 256x32, B fragments reused by two MFMAs
    DX2(2, 1, 0, 2, 1, 1)   // 128x32 with two waves
    DX2(2, 1, 0, 2, 1, 2)
    DX2(2, 1, 0, 4, 1, 1)   // 256x32 with two waves
    DX2(2, 1, 0, 1, 1, 1)   // 64x32
    DX2(2, 1, 0, 1, 1, 2)
    DX2(1, 1, 0, 2, 1, 1)   // 64x32 one wave
    DX2(1, 1, 0, 4, 1, 1)   // 128x32 one wave
    DX2(1, 1, 0, 4, 1, 2)
    // round 4: the four classes as 128 columns of ONE GEMM (DxC2MP): dY staged once, checksums must equal the per-class kernels'
#define DX2M(WM, WN, TM, TN, T) { using P = DxC2MP<WM, WN, TM, TN>; \
    CK(hipMemset(dx1, 0, n1 * 4)); run<P, T>("dx_c2 merged w" #WM "x" #WN " t" #TM "x" #TN " teams" #T, dim3(m_tiles<P>(d2.M) * (128 / (WN * TN * 32)), 1, 1), d2, dx1, n1); }
    DX2M(2, 2, 1, 2, 1)     // 64 x 128, 400 workgroups
    DX2M(2, 2, 1, 2, 2)
    DX2M(2, 2, 2, 2, 1)     // 128 x 128, 200
    DX2M(2, 2, 2, 2, 2)
    DX2M(1, 4, 1, 1, 1)     // 32 x 128, 800
    DX2M(1, 4, 2, 1, 1)     // 64 x 128, waves own a class
    DX2M(1, 4, 2, 1, 2)
    DX2M(2, 4, 1, 1, 1)     // 64 x 128 with 8 waves
    DX2M(4, 2, 1, 2, 1)     // 128 x 128 with 8 waves
    DX2M(2, 2, 1, 1, 1)     // 64 x 64: two column tiles (dY staged twice)
    DX2M(2, 2, 1, 1, 2)
    DX2M(4, 1, 1, 2, 1)     // 128 x 64
    DX2M(1, 2, 1, 2, 1)     // 32 x 128 with two waves
    DX2M(1, 2, 1, 2, 2)
    DX2M(1, 4, 1, 1, 2)     // 32 x 128, two teams
    DX2M(4, 4, 1, 1, 1)     // 128 x 128 with 16 waves
    DX2M(2, 4, 2, 1, 1)     // 128 x 128 with 8 waves, waves own a class
    return 0;
}
